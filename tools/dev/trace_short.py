"""Per kernel of a rocprofv3 kernel trace: average us, launches, grid / workgroup size, LDS, VGPRs."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    k = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0])
    a[0] += d; a[1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    wgs = int(k[1]) * int(k[2]) * int(k[3]) // max(1, int(k[4]))
    print("%7.1f us x%-3d wgs %-5d threads %-4s lds %-6s vgpr %-3s %s" % (t / c / 1e3, c, wgs, k[4], k[5], k[6], k[0]))

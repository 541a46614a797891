"""Every kernel family of a model's training step against its bound: reduce the passes of tools/family_roofline.sh —
a rocprofv3 kernel trace (time per launch, steady-state window: the last quarter of the dispatches, as
tools/model_prof_report.py), separate --pmc FETCH_SIZE / WRITE_SIZE passes (HBM bytes per launch; FETCH doubled: the gfx950
note of MI355X_MICROARCH.md) and a SQ_VALU_MFMA_BUSY_CYCLES pass (matrix-pipe share) — to one line per family and per
kernel:  share of the step | us per launch | measured HBM MB per launch | TB/s = fraction of the 8 TB/s roofline |
matrix-pipe busy share.  HBM-bound families (norms, lattice, raster, elementwise) read their fraction in the TB/s column,
the matrix-core families (pointwise GEMMs, grouped conv) in the busy column.
    python3 tools/family_roofline.py <dir with trace/ fetch/ write/ mfma/> [title] [kernels per family]"""
import collections
import csv
import glob
import sys

from model_prof_report import family

HBM_PEAK = 8.0e12


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def counters(root, tag):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    f = one(root + "/" + tag + "/**/*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def mean_tail(v):
    v = v[len(v) // 4:] if len(v) >= 8 else v
    return sum(v) / max(1, len(v))


def main():
    root = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else root
    ntop = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
            for r in csv.DictReader(open(one(root + "/trace/**/*kernel_trace.csv")))]
    rows.sort()
    rows = rows[int(0.75 * len(rows)):]
    fetch, write, mfma = counters(root, "fetch"), counters(root, "write"), counters(root, "mfma")
    per = collections.defaultdict(lambda: [0.0, 0])
    for a, b, name in rows:
        per[name][0] += b - a
        per[name][1] += 1
    total = sum(v[0] for v in per.values())
    fam = collections.defaultdict(lambda: {"ns": 0.0, "bytes": 0.0, "busy": 0.0, "kernels": []})
    for name, (ns, calls) in per.items():
        by = (2 * mean_tail(fetch[name].get("FETCH_SIZE", [0.0])) + mean_tail(write[name].get("WRITE_SIZE", [0.0]))) * 1024
        busy = mean_tail(mfma[name].get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])) / 4 / 256       # cycles per CU's four pipes
        f = fam[family(name)]
        f["ns"] += ns
        f["bytes"] += by * calls
        f["busy"] += busy * calls
        f["kernels"].append((ns, calls, by, busy, name))
    print("== %s: %.2f ms of kernel time in the steady-state window; HBM roofline %.0f TB/s, matrix pipes at the 2.4 GHz clock"
          % (title, total / 1e6, HBM_PEAK / 1e12))
    print("   share | family | measured HBM traffic / time = TB/s (fraction of the roofline) | matrix pipes busy")
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
        tbs = f["bytes"] / max(f["ns"], 1.0) * 1e9 / 1e12
        busy = f["busy"] / (f["ns"] * 2.4)          # busy cycles / (ns * 2.4 cycles per ns)
        print("  %5.1f %%  %-62s %5.2f TB/s (%.2f)   mfma %4.1f %%" % (100 * f["ns"] / total, name[:62], tbs, tbs * 1e12 / HBM_PEAK, 100 * busy))
        for ns, calls, by, bz, kn in sorted(f["kernels"], reverse=True)[:ntop]:
            us = ns / calls / 1e3
            short = kn.replace("(anonymous namespace)::", "").replace("void ", "")[:84]
            print("           %5.1f %%  x%-5d %7.1f us  %7.1f MB  %5.2f TB/s (%.2f)  mfma %4.1f %%  %s"
                  % (100 * ns / total, calls, us, by / 1e6, by / us / 1e6, by / us / 1e6 * 1e12 / HBM_PEAK, 100 * bz / (us * 2.4e3), short))


if __name__ == "__main__":
    main()

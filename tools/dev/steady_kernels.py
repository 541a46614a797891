"""Reduce a rocprofv3 kernel trace to the steady state: per-kernel totals over the last WINDOW ms of the run.
usage: steady_kernels.py <kernel_trace.csv> [window_ms] [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - win * 1e6
agg, cnt = collections.Counter(), collections.Counter()
for r in rows:
    s = int(r["Start_Timestamp"])
    if s >= cut:
        n = r["Kernel_Name"]
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = n[:100]
        agg[n] += int(r["End_Timestamp"]) - s
        cnt[n] += 1
tot = sum(agg.values())
print("window %.1f ms, kernels busy %.2f ms, %d launches" % (win, tot / 1e6, sum(cnt.values())))
for n, d in agg.most_common(top):
    print("%9.1f us %5.1f%% %5d  %s" % (d / 1e3, 100.0 * d / tot, cnt[n], n))

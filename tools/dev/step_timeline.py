"""tools/dev/step_timeline.py <kernel_trace.csv> [fraction=0.25]: how busy the GPU is inside a model step's graph replays — the last
`fraction` of the dispatches of a tools/*_step_bench.py run under rocprofv3 --kernel-trace: wall time of the window, time with at
least one kernel running (union of the kernel intervals), the sum of kernel time (concurrency = sum / union), the idle gaps between
kernels (count, total, the largest with the kernels on either side) and the kernels that run ALONE the longest."""
import csv, sys, collections
path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
rows = rows[int((1 - frac) * len(rows)):]
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for a, b, n in rows:
    ev.append((a, 1, n)); ev.append((b, -1, n))
ev.sort(key=lambda e: (e[0], -e[1]))
busy = 0; depth = 0; last = t0; alone = collections.defaultdict(int); gaps = []; cur = []
prev_end_name = None
for t, d, n in ev:
    if depth > 0: busy += t - last
    if depth == 1 and cur: alone[short(cur[0])] += t - last
    if depth == 0 and t > last and prev_end_name is not None: gaps.append((t - last, prev_end_name, short(n)))
    if d == 1: cur.append(n)
    else:
        cur.remove(n); prev_end_name = short(n)
    depth += d; last = t
wall = t1 - t0
ksum = sum(b - a for a, b, _ in rows)
print("window: %d kernels, wall %.2f ms, some kernel running %.2f ms (%.1f %%), kernel time %.2f ms (concurrency %.2f)" % (
    len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, ksum / 1e6, ksum / busy))
gaps.sort(reverse=True)
tot = sum(g[0] for g in gaps)
print("idle gaps: %d, total %.2f ms (%.1f %% of the wall); > 5 us: %d totalling %.2f ms" % (
    len(gaps), tot / 1e6, 100.0 * tot / wall, sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
for g in gaps[:12]:
    print("   %7.1f us idle between %s  ->  %s" % (g[0] / 1e3, g[1], g[2]))
print("running alone (no other kernel on the device), top 12:")
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print("   %7.2f ms  %s" % (v / 1e6, n))
# the neighbourhood of the largest gaps: who ran last before them, who ran first after them (queue ids show stream hand-offs)
if len(sys.argv) > 3:
    full = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(path))]
    full.sort()
    full = full[int((1 - frac) * len(full)):]
    ends = sorted(full, key=lambda r: r[1])
    big = sorted(((full[i + 1][0] - max(x[1] for x in full[:i + 1]), i) for i in range(len(full) - 1)), reverse=True)[:int(sys.argv[3])]
    for gap, i in big:
        if gap <= 0: continue
        print("gap %.1f us:" % (gap / 1e3))
        for a, b, n, q in full[max(0, i - 3):i + 1]:
            print("     before  q%s  %9.1f .. %9.1f us  %s" % (q, (a - t0) / 1e3, (b - t0) / 1e3, short(n)))
        for a, b, n, q in full[i + 1:i + 4]:
            print("     after   q%s  %9.1f .. %9.1f us  %s" % (q, (a - t0) / 1e3, (b - t0) / 1e3, short(n)))

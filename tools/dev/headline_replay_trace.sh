#!/bin/bash
# Per-kernel durations INSIDE the headline's HIP-graph replays (second half): rocprofv3 --kernel-trace over bench.py,
# reduced over the dispatches of the timed region (the last 4 x steps launches of the run: bench.py times its eager per-pass loops
# first, then replays the graph) beside the eager per-pass loops of the same run.   -> gpurun_out/headline_replay_trace.txt
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out
rm -rf $OUT/prof_hl; mkdir -p $OUT/prof_hl
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_hl -o h -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/prof_hl/bench.log 2>&1)
python3 - "$OUT" <<'PY' > $OUT/headline_replay_trace.txt
import csv, glob, sys, json, collections, statistics
out = sys.argv[1]
f = glob.glob(out + "/prof_hl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
short = lambda n: n.replace("void (anonymous namespace)::", "").split("(")[0]
ours = [(a, b, short(n)) for a, b, n in rows if any(k in n for k in ("scatter_quad", "gather_ci", "slice_bwd", "splat_max_bwd"))]
steps = 200
replay = ours[-4 * steps:]            # the timed region: the last 20 replays of the 10-step graph
nloop = len(ours) - 4 * (steps + 20 + 10 + 1 + 1)   # in front of: warm-up replays, the multi-graph's set-up replay, the single graph's, the eager step
eager = ours[4:max(4, nloop)]
def table(rs, title):
    acc = collections.defaultdict(list)
    for a, b, n in rs:
        acc[n].append(b - a)
    print(title)
    tot = 0.0
    for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        avg = sum(v) / len(v)
        tot += avg
        print("  %-62s %5d launches  avg %7.2f us  median %7.2f  min %7.2f  max %7.2f" % (n[:62], len(v), avg / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3))
    print("  sum of the averages %.2f us" % (tot / 1e3))
table(replay, "== inside the graph replays of the timed region (last %d dispatches = %d steps)" % (len(replay), steps))
walls = [(replay[k + 39][1] - replay[k][0]) / 1e4 for k in range(0, len(replay), 40)]
print("  wall per step inside a 10-step graph (first start -> last end of its 40 kernels): median %.2f us (min %.2f, max %.2f)" % (statistics.median(walls), min(walls), max(walls)))
gaps = [replay[i + 1][0] - replay[i][1] for i in range(len(replay) - 1) if (i + 1) % 40]
print("  gap between consecutive kernels of a graph: median %.2f us" % (statistics.median(gaps) / 1e3))
table(eager, "== eager per-pass loops of the same run (bench.py time_passes: the SAME kernel back to back)")
for line in open(out + "/prof_hl/bench.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print("bench line of this (profiled) run: %.1f M points/s, %.4f ms per step; passes_ms %s" % (d["value"] / 1e6, d["ms_per_step"], {k: round(v, 4) for k, v in d["passes_ms"].items()}))
PY
cat $OUT/headline_replay_trace.txt
python3 $R/bench.py --no-cpu-baseline > $OUT/r6_bench_line_a.json 2> $OUT/r6_bench_line_a.err; tail -c 600 $OUT/r6_bench_line_a.json

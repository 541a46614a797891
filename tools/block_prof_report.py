"""Summarise gpurun_out/prof_block/*kernel_stats.csv of tools/block_prof.py: µs per iteration per kernel,
MIOpen find-mode kernels (run once while it picks algorithms) left out."""
import csv
import glob
import sys

ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_block/*kernel_stats.csv")[0])))
tot = 0.0
out = []
for r in rows:
    n = r["Name"]
    calls = int(r["Calls"])
    if calls < ITERS // 2:          # find-mode / one-off kernels
        continue
    us = float(r["TotalDurationNs"]) / ITERS / 1e3
    tot += us
    out.append((us, calls / ITERS, n))
out.sort(reverse=True)
print(f"steady-state kernel time per iteration: {tot:.0f} us")
for us, c, n in out[: int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print(f"{us:8.1f} us {c:5.1f} calls  {n[:120]}")

#!/bin/bash
# The headline over the seeds the ranks of an 8-GPU run draw (bench.py: seed + rank): step time, Splat(max) backward time and exact
# ties per seed on ONE GPU (run through gpurun from the repo root); the summary goes to profiles/rN_bench_seeds.txt
for s in 1234 1235 1236 1237 1238 1239 1240 1241; do
  python bench.py --seed $s --no-cpu-baseline --steps 200 --warmup 50 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('seed %d  ms_per_step %.5f  value %.1f M points/s  ties_seen %d  passes_us %s' % ($s, r['ms_per_step'], r['value']/1e6, r['ties_seen'], {k: round(v*1e3,1) for k,v in r['passes_ms'].items()}))"
done

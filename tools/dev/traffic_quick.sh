#!/bin/bash
# quick FETCH_SIZE / WRITE_SIZE per kernel of the bench step (two short PMC passes); prints MB per launch
R=/root/repo; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
A="--steps 10 --warmup 2 --no-cpu-baseline --no-graph $@"
rm -rf $OUT/q_fetch $OUT/q_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/q_fetch -- python3 $R/bench.py $A > /dev/null 2> $OUT/q_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/q_write -- python3 $R/bench.py $A > /dev/null 2> $OUT/q_write.err
cd $R
python3 - <<'PY'
import csv, glob, collections
def avg(pat, ctr):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}
fe = avg("gpurun_out/q_fetch/*/*counter_collection.csv", "FETCH_SIZE")
wr = avg("gpurun_out/q_write/*/*counter_collection.csv", "WRITE_SIZE")
for k in sorted(set(fe) | set(wr)):
    if "at::" in k or "Cijk" in k: continue
    print("%-55s fetch %7.1f MB (x2 gfx950)  write %7.1f MB  total %7.1f MB" % (k[:55], 2 * fe.get(k, 0) * 1.024e-3, wr.get(k, 0) * 1.024e-3, (2 * fe.get(k, 0) + wr.get(k, 0)) * 1.024e-3))
PY

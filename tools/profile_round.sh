#!/bin/bash
# Collect a round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root: bash tools/profile_round.sh):
#   1. --kernel-trace --stats of the default bench command
#   2. --pmc FETCH_SIZE   (separate pass)
#   3. --pmc WRITE_SIZE   (separate pass)
# Outputs land in gpurun_out/prof_*; tools/pmc_traffic.py reduces them into profiles/.
set -u
R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $R/bench.py $ARGS > $OUT/prof_stats_bench.json 2> $OUT/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $R/bench.py $ARGS --no-graph > $OUT/prof_fetch_bench.json 2> $OUT/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 $R/bench.py $ARGS --no-graph > $OUT/prof_write_bench.json 2> $OUT/prof_write.err
cd $R
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head -20
# reduce on the box too (the CSVs are large): gpurun_out/prof_kernel_stats.csv, gpurun_out/prof_traffic.json
S=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
F=$(find gpurun_out/prof_fetch -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/prof_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$S" "$F" "$W" gpurun_out/prof > gpurun_out/prof_traffic_stdout.json
cp "$S" gpurun_out/prof_rocprofv3_kernel_stats_full.csv

#!/bin/bash
# Kernel-level breakdown of one MultiHeadUnion fwd+bwd per zoo stage and of the three model training steps
# (run through gpurun from the repo root: bash tools/block_prof.sh) -> gpurun_out/block_prof_report.txt
set -u
R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
: > $OUT/block_prof_report.txt
for stage in 1 2 3; do
  rm -rf $OUT/prof_block
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_block -o b -- python3 $R/tools/block_prof.py $stage > $OUT/prof_block_$stage.log 2>&1)
  echo "== MultiHeadUnion stage $stage fwd+bwd, B8 N4096 (tools/block_prof.py $stage, 100 iterations, eager launches)" >> $OUT/block_prof_report.txt
  python3 tools/block_prof_report.py 30 100 >> $OUT/block_prof_report.txt 2>&1
done
rm -rf $OUT/prof_block

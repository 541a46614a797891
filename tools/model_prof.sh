#!/bin/bash
# Kernel-family breakdown of the three model training steps (run through gpurun from the repo root: bash tools/model_prof.sh)
#   -> gpurun_out/model_breakdown.txt   (rocprofv3 --kernel-trace --stats over tools/{segmenter,classifier,inpainter}_step_bench.py)
set -u
R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
: > $OUT/model_breakdown.txt
for m in segmenter classifier inpainter; do
  rm -rf $OUT/prof_model
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_model -o m -- python3 $R/tools/${m}_step_bench.py > $OUT/prof_model_$m.log 2>&1)
  grep -h "training step" $OUT/prof_model_$m.log >> $OUT/model_breakdown.txt
  S=$(find $OUT/prof_model -name "*kernel_trace.csv" | head -1)
  python3 tools/model_prof_report.py "$S" "$m training step (graph replays)" ${NTOP:-5} >> $OUT/model_breakdown.txt 2>&1
  echo >> $OUT/model_breakdown.txt
done
rm -rf $OUT/prof_model
tail -30 $OUT/model_breakdown.txt

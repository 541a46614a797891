#!/bin/bash
# The plane-resident MHCT core INSIDE the training steps it ships in (run through gpurun from the repo root):
# graphed segmenter and classifier steps with CLOUDCT_FUSED_CORE = 0 / 1 x CLOUDCT_HEAD_STREAMS = 0 / auto (the two heads of a
# union on forked streams under capture), then rocprofv3 kernel traces of the REPLAYS with the core on: the in-step duration of
# mhct_core_fwd_kernel with and without a sibling stream.   -> gpurun_out/fused_core_in_step.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
REP=$OUT/fused_core_in_step.txt
: > $REP
for m in segmenter classifier; do
  for fc in 1 0; do
    for hs in auto 0; do
      line=$(cd $R && CLOUDCT_FUSED_CORE=$fc CLOUDCT_HEAD_STREAMS=$hs python3 tools/${m}_step_bench.py 2>&1 | grep "training step" | sed 's/.*fp32: //')
      echo "$m  FUSED_CORE=$fc HEAD_STREAMS=$hs : $line" >> $REP
    done
  done
done
echo >> $REP
for m in segmenter classifier; do
  for hs in auto 0; do
    rm -rf $OUT/prof_fc
    (cd /tmp && TMPDIR=/tmp CLOUDCT_FUSED_CORE=1 CLOUDCT_HEAD_STREAMS=$hs rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_fc -o m -- python3 $R/tools/${m}_step_bench.py > $OUT/prof_fc.log 2>&1)
    S=$(find $OUT/prof_fc -name "*kernel_trace.csv" | head -1)
    echo "== $m step, FUSED_CORE=1 HEAD_STREAMS=$hs: kernels of the graph replays (last quarter of the dispatches), raster family" >> $REP
    python3 $R/tools/model_prof_report.py "$S" "$m" 12 2>&1 | grep -i "mhct_core\|raster\|scatter_quad\|gather_ci\|slice_bwd\|splat_max\|wall\|kernel time" >> $REP
    echo >> $REP
  done
done
rm -rf $OUT/prof_fc
cat $REP

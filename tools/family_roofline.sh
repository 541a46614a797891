#!/bin/bash
# Kernel families of a model training step against their bounds (run through gpurun from the repo root:
#   bash tools/family_roofline.sh segmenter|classifier|inpainter): kernel trace + FETCH_SIZE + WRITE_SIZE + MFMA-busy passes, each
# its own rocprofv3 run (no --pmc together with tracing other than kernel), reduced by tools/family_roofline.py
#   -> gpurun_out/family_roofline_<model>.txt
set -u
R=/root/repo
M=${1:-segmenter}
OUT=$R/gpurun_out/famroof_$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/tools/${M}_step_bench.py > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/tools/${M}_step_bench.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/tools/${M}_step_bench.py > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/mfma -o m -- python3 $R/tools/${M}_step_bench.py > $OUT/mfma.log 2>&1
cd $R/tools
python3 family_roofline.py $OUT "$M training step" ${NTOP:-4} > $R/gpurun_out/family_roofline_$M.txt 2>&1
rm -rf $OUT/*/*.csv $OUT/*/*/*.csv 2>/dev/null
cat $R/gpurun_out/family_roofline_$M.txt

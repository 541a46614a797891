#!/bin/bash
# MFMA utilisation of the grouped-conv kernels from hardware counters (run through gpurun from the repo root):
#   pass 1: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA   pass 2: --kernel-trace --stats   (separate runs)
# tools/mfma_util.py reduces them to profiles/r1_gconv_mfma_util.txt
set -u
R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in 2d 3d 8c32 32c4 8c64 4c64 2c64 8c64_2d; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/mfma_pmc_$cfg -o m -- python3 $R/tools/gconv_prof.py $cfg > /dev/null 2> $OUT/mfma_pmc_$cfg.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mfma_stats_$cfg -o m -- python3 $R/tools/gconv_prof.py $cfg > /dev/null 2> $OUT/mfma_stats_$cfg.err
done
cd $R
python3 tools/mfma_util.py gpurun_out/gconv_mfma_util.txt

#!/bin/bash
# rocprofv3 passes over any python program (run through gpurun from the repo root: bash tools/kprof.sh TAG script.py [args]):
# kernel trace, two SQ counter passes (incl. the matrix-core busy cycles), FETCH_SIZE, WRITE_SIZE — each its own run (no --pmc
# with tracing other than kernel).  Reduced by tools/kprof_report.py -> gpurun_out/kprof_TAG.txt (per kernel, per launch)
set -u
R=/root/repo
TAG=$1; shift
OUT=$R/gpurun_out/kprof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/"$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT/sq_a -o a -- python3 $R/"$@" > $OUT/sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVES \
  --output-format csv -d $OUT/sq_b -o b -- python3 $R/"$@" > $OUT/sq_b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq_c -o c -- python3 $R/"$@" > $OUT/sq_c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/"$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/"$@" > $OUT/write.log 2>&1
cd $R
python3 tools/kprof_report.py $OUT > gpurun_out/kprof_$TAG.txt 2>&1
rm -rf $OUT/*/*.csv $OUT/*/*/*.csv 2>/dev/null
cat gpurun_out/kprof_$TAG.txt

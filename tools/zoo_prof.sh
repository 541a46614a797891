#!/bin/bash
# rocprofv3 passes over the zoo head shapes (run through gpurun from the repo root: bash tools/zoo_prof.sh [configs...]):
# kernel trace, two SQ counter passes, FETCH_SIZE, WRITE_SIZE — each its own run (no --pmc with tracing other than kernel).
# Reduced on the box by tools/zoo_prof_report.py -> gpurun_out/zoo_prof_report.txt
set -u
R=/root/repo
OUT=$R/gpurun_out/zoo_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/tools/zoo_prof.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT/sq_a -o a -- python3 $R/tools/zoo_prof.py "$@" > $OUT/sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES \
  --output-format csv -d $OUT/sq_b -o b -- python3 $R/tools/zoo_prof.py "$@" > $OUT/sq_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/tools/zoo_prof.py "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/tools/zoo_prof.py "$@" > $OUT/write.log 2>&1
cd $R
python3 tools/zoo_prof_report.py gpurun_out/zoo_prof "$@" > gpurun_out/zoo_prof_report.txt 2>&1
tail -5 gpurun_out/zoo_prof_report.txt

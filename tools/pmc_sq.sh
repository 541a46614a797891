#!/bin/bash
# SQ counter passes over one command (default: the bench without graph replay); run through gpurun from the repo root:
#   tools/pmc_sq.sh [python3 script args...]        -> gpurun_out/pmc_sq_{a,b}/*.csv ; reduce with tools/pmc_sq.py
set -u
R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT/pmc_sq_a -o a -- "$@" > /dev/null 2> $OUT/pmc_sq_a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES \
  --output-format csv -d $OUT/pmc_sq_b -o b -- "$@" > /dev/null 2> $OUT/pmc_sq_b.err
cd $R
find gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b -name "*counter_collection.csv"

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sorted3_gpu.py -x -q > gpurun_out/r6_t_sorted3.txt 2>&1
tail -15 gpurun_out/r6_t_sorted3.txt
timeout 900 python -m pytest tests/test_tickets_gpu.py tests/test_sorted_gpu.py -x -q > gpurun_out/r6_t_reg.txt 2>&1
tail -5 gpurun_out/r6_t_reg.txt
bash tools/dev/zoo_ab.sh s3pf

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sorted3_gpu.py tests/test_tickets_gpu.py tests/test_sorted_gpu.py tests/test_raster_gpu.py tests/test_tie_rule_gpu.py -x -q > gpurun_out/r6_t_a.txt 2>&1
tail -5 gpurun_out/r6_t_a.txt
timeout 1500 python -m pytest tests/test_headline_gpu.py -x -q > gpurun_out/r6_t_b.txt 2>&1
tail -5 gpurun_out/r6_t_b.txt
python tools/zoo_sweep.py --tickets-only > gpurun_out/r6_zoo_sweep_b.txt 2>&1
cat gpurun_out/r6_zoo_sweep_b.txt

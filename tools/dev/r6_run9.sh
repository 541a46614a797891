#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_sorted3_gpu.py tests/test_sorted_gpu.py tests/test_tickets_gpu.py tests/test_raster_gpu.py tests/test_tie_rule_gpu.py -x -q 2>&1 | tail -4
python tools/zoo_sweep.py --tickets-only > gpurun_out/r6_zoo_sweep_c.txt 2>&1
grep tickets gpurun_out/r6_zoo_sweep_c.txt

#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_sorted3_gpu.py tests/test_sorted_gpu.py tests/test_tickets_gpu.py -x -q 2>&1 | tail -6
P="python tools/dev/zoo_shape.py"
for cfg in "16 16 2 8 4096" "16 16 2 8 2048" "16 16 2 2 16384"; do
  $P $cfg 2>&1 | grep -v amdgpu
  CLOUDCT_SORTED2S=0 $P $cfg 2>&1 | grep -v amdgpu | head -1
done

#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_sorted3_gpu.py -x -q 2>&1 | tail -4
P="python tools/dev/zoo_shape.py"
$P 32 8 3 8 2048 sum 2>&1 | grep -v amdgpu
CLOUDCT_SORTED=0 $P 32 8 3 8 2048 sum 2>&1 | grep -v amdgpu
$P 32 8 3 8 1024 sum 2>&1 | grep -v amdgpu
CLOUDCT_SORTED=0 $P 32 8 3 8 1024 sum 2>&1 | grep -v amdgpu

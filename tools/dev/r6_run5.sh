#!/bin/bash
cd "$(dirname "$0")/../.."
python tools/dev/zoo_shape.py 16 64 2 8 4096 2>&1 | tail -8
CLOUDCT_WIDE=0 python tools/dev/zoo_shape.py 16 64 2 8 4096 2>&1 | tail -8
timeout 600 python -m pytest tests/test_sorted3_gpu.py -x -q 2>&1 | tail -4
python tools/dev/zoo_shape.py 32 8 3 8 4096 2>&1 | tail -4
python tools/dev/zoo_shape.py 32 8 3 8 2048 2>&1 | tail -4
python tools/dev/zoo_shape.py 32 8 3 2 16384 2>&1 | tail -4

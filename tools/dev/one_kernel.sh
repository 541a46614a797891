#!/bin/bash
# tools/dev/one_kernel.sh 'kernel<args>' [extra hipcc flags]: compiles that kernel alone, prints its registers / spills / scratch,
# leaves the ISA in /tmp/one.s
K=$1; shift
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I include -I cloud_transformers_amd/csrc "-DCT_ONE=$K" "$@" \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj -c tools/dev/one_kernel.hip -o /tmp/one.o 2> /tmp/one.txt || { tail -30 /tmp/one.txt; exit 1; }
python tools/dev/regs_report.py /tmp/one.txt
ls /tmp/one*.s 2>/dev/null | head -3

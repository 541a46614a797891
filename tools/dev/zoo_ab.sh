#!/bin/bash
# tools/dev/zoo_ab.sh NAME...: tools/zoo_sweep.py --tickets-only under the product library and under each experimental
# library cloud_transformers_amd/lib/libcloudct_NAME.so, one after the other on this box -> gpurun_out/zoo_ab_<NAME>.txt
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python tools/zoo_sweep.py --tickets-only > gpurun_out/zoo_ab_base.txt 2>&1
for n in "$@"; do
  CLOUDCT_LIB=$PWD/cloud_transformers_amd/lib/libcloudct_$n.so python tools/zoo_sweep.py --tickets-only > gpurun_out/zoo_ab_$n.txt 2>&1
done
python tools/zoo_sweep.py --tickets-only > gpurun_out/zoo_ab_base2.txt 2>&1
tail -n 20 gpurun_out/zoo_ab_*.txt

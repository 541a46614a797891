#!/bin/bash
# tools/dev/build_raster_exp.sh NAME [extra -D flags]: recompiles ct_raster.hip alone with the flags and links it with the product's
# other objects into cloud_transformers_amd/lib/libcloudct_NAME.so (A/B raster kernels via CLOUDCT_LIB=...)
set -e
N=$1; shift
cd /root/repo/cloud_transformers_amd/csrc
mkdir -p /tmp/rexp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off "$@" -I /root/repo/include -c ct_raster.hip -o /tmp/rexp/ct_raster_$N.o
objs=$(ls ../lib/obj/*.o | grep -v ct_raster.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/rexp/ct_raster_$N.o $objs -o ../lib/libcloudct_$N.so
echo built ../lib/libcloudct_$N.so

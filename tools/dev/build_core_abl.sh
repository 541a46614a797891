#!/bin/bash
# tools/dev/build_core_abl.sh NAME [-D...]: lib/libcloudct_NAME.so = the library with ct_mhct.hip rebuilt with the given defines
# (the other objects come from lib/obj/, built by _lib.build()).  Select with CLOUDCT_LIB=... for A/B runs.
N=$1; shift
cd /root/repo/cloud_transformers_amd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off "$@" -I /root/repo/include -c csrc/ct_mhct.hip -o lib/obj/ct_mhct_$N.o || exit 1
OBJS=$(ls lib/obj/*.o | grep -v ct_mhct)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS lib/obj/ct_mhct_$N.o -o lib/libcloudct_$N.so

#!/bin/bash
# tools/dev/build_exp.sh N [extra flags]: builds cloud_transformers_amd/lib/libcloudct_expN.so with -DCT_EXP=N (A/B kernels via CLOUDCT_LIB)
N=$1; shift
cd /root/repo/cloud_transformers_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -DCT_EXP=$N "$@" -I /root/repo/include \
  ct_raster.hip ct_lattice.hip ct_gconv.hip ct_chamfer.hip ct_emd.hip ct_adain.hip ct_bnorm.hip ct_mhct.hip ct_pwgemm.hip -o ../lib/libcloudct_exp$N.so

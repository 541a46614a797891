#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out
CLOUDCT_LIB=$R/cloud_transformers_amd/lib/libcloudct_s3st.so python tools/dev/s3_stamps.py > gpurun_out/r6_s3_stamps.txt 2>&1
CLOUDCT_LIB=$R/cloud_transformers_amd/lib/libcloudct_s3st.so python tools/dev/s3_stamps.py 32 8 8 2048 >> gpurun_out/r6_s3_stamps.txt 2>&1
cat gpurun_out/r6_s3_stamps.txt

# A/B of a development build against the product library on the same box: bash tools/dev/pw_ab.sh N  (libcloudct_expN.so)
for r in 1 2; do
  echo "== product"; python tools/pw_gemm_bench.py 8,848,512,4096 8,592,512,4096 2,512,512,16384 2>&1 | grep -E "fwd|dgrad|wgrad" | awk -F'|' '{print $1, $2, $3}'
  echo "== exp$1"; CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_exp$1.so python tools/pw_gemm_bench.py 8,848,512,4096 8,592,512,4096 2,512,512,16384 2>&1 | grep -E "fwd|dgrad|wgrad" | awk -F'|' '{print $1, $2, $3}'
done

for k in 0 1 2 3 4; do
  if [ $k = 0 ]; then unset CLOUDCT_LIB; else export CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_exp4$k.so; fi
  echo "== ablation $k"; python tools/pw_gemm_bench.py 8,848,512,4096 2>&1 | grep -E "fwd|dgrad|wgrad" | awk -F'|' '{print $2, $3}'
done

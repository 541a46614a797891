# Ablations of pw_gemm_kernel on one box (outputs are wrong by design; only the times matter).  Build first:
#   for k in 1 3 4; do bash tools/dev/build_exp.sh 4$k -DPW_ABL=$k; done
#   1: no f16 split in pw_split8 (raw bits to LDS)   3: no global loads in the main loop   4: no output stores
for k in 0 3 4; do
  if [ $k = 0 ]; then unset CLOUDCT_LIB; else export CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_exp4$k.so; fi
  echo "== ablation $k"; python tools/pw_gemm_bench.py 8,848,512,4096 2>&1 | grep -E "fwd|dgrad|wgrad" | awk -F'|' '{print $2, $3}'
done

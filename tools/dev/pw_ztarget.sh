for lib in "" 81 82; do
  if [ -z "$lib" ]; then unset CLOUDCT_LIB; else export CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_exp$lib.so; fi
  echo "== ztarget lib '$lib'"
  python tools/pw_gemm_bench.py 8,848,512,4096 8,592,512,4096 8,208,512,4096 8,512,512,4096 2>&1 | grep wgrad | awk -F'|' '{print $1, $3}'
  python tools/segmenter_step_bench.py 2>&1 | grep "training step" | cut -c150-300
done

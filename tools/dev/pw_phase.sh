for l in "" "$@"; do      # the product library, then the experimental ones named on the command line (tools/dev/build_exp.sh)
  if [ -z "$l" ]; then unset CLOUDCT_LIB; else export CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_$l.so; fi
  echo "== $l"; python tools/pw_gemm_bench.py 8,848,512,4096 2>&1 | grep -E "fwd|dgrad|wgrad" | awk -F'|' '{print $2, $3}'
done

# two bidders per lane group (default) against one (-DCT_EMD_PAIR_FROM=1000000: bash tools/dev/build_exp.sh 70 -DCT_EMD_PAIR_FROM=1000000) on one box
for l in "" exp70; do
  if [ -z "$l" ]; then unset CLOUDCT_LIB; echo "== pairs"; else export CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_$l.so; echo "== single"; fi
  python tools/dev/emd_inpainter_bidders.py 2>&1 | grep "50 iterations\|oracle"
  python tools/loss_bench.py 2>&1 | grep -i emd
done

#!/bin/bash
# A/B of the plane-resident MHCT core (ct_mhct_core_fwd) against the three-kernel chain (run through gpurun from the repo root:
# bash tools/fused_core_ab.sh): op level (tools/core_bench.py, HIP events), then the stage-3 zoo block and the headline-shaped
# block fwd+bwd under rocprofv3 --kernel-trace --stats with the dispatch on and off  ->  gpurun_out/fused_core_ab.txt
set -u
R=/root/repo
OUT=$R/gpurun_out
mkdir -p $OUT
REP=$OUT/fused_core_ab.txt
{
echo "== op level: forward of the core, HIP events, 200 launches (tools/core_bench.py)"
python3 tools/core_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo
echo "== block level: MultiHeadUnion stage 3 = (C16,16^2)+(C32,8^3), B8 N4096, fwd+bwd graphed (tools/zoo_blocks_bench.py)"
for f in 1 0; do echo "CLOUDCT_FUSED_CORE=$f"; CLOUDCT_FUSED_CORE=$f python3 tools/zoo_blocks_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "stage 3\|12-block"; done
} > $REP 2>&1
for f in 1 0; do
  rm -rf $OUT/prof_block
  (cd /tmp && TMPDIR=/tmp CLOUDCT_FUSED_CORE=$f rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_block -o b -- python3 $R/tools/block_prof.py 3 > $OUT/prof_block_ab_$f.log 2>&1)
  echo >> $REP
  echo "== rocprofv3 --kernel-trace --stats, MultiHeadUnion stage 3 fwd+bwd B8 N4096 (tools/block_prof.py 3, 100 iterations), CLOUDCT_FUSED_CORE=$f: raster / conv / core kernels" >> $REP
  python3 tools/block_prof_report.py 60 100 | grep -i "steady\|mhct\|scatter\|gather\|slice_\|splat_\|gconv\|occupancy\|sum_parts\|quad_kernel" >> $REP 2>&1
done
rm -rf $OUT/prof_block
tail -40 $REP

#!/bin/bash
# Regenerate profiles/rN_tools_output.txt on the GPU box: every benchmark tool quoted in DESIGN.md, in one file.
# Usage (through gpurun, from the repo root): bash tools/run_all_benches.sh > gpurun_out/tools_output.txt
for t in block_bench block_graph_bench zoo_blocks_bench adain_block_bench segmenter_step_bench classifier_step_bench inpainter_step_bench gconv_bench adain_bench bn_bench \
         zoo_sweep loss_bench conv1d_bench wrw_time gconv_fwd_time gconv64_bench; do
  echo "== tools/$t.py"
  python tools/$t.py 2>&1 | grep -v "amdgpu.ids\|Warning\|^  \|Consider using" | grep -v "^$"
done

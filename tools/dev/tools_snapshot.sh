# One file with the output of every bench tool at HEAD: bash tools/dev/tools_snapshot.sh > gpurun_out/tools_output.txt (through gpurun)
for t in block_bench block_graph_bench zoo_blocks_bench adain_block_bench segmenter_step_bench classifier_step_bench inpainter_step_bench \
         gconv_bench adain_bench bn_bench zoo_sweep loss_bench conv1d_bench pw_gemm_bench gconv64_bench; do
  echo "== tools/$t.py"
  timeout 600 python tools/$t.py 2>&1 | grep -v -E "amdgpu.ids|UserWarning|_warn_once|Consider using tensor.detach|python_variable_methods|RCCL version|HIP version|ROCm version|Hostname|Librccl path|^  f\""
done

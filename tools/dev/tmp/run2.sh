timeout 900 python -m pytest tests/test_mhct_core_gpu.py -x -q 2>&1 | tail -8
echo "== full"; timeout 300 python tools/core_bench.py --S=0 --S=1 --S=2 --S=4 2>&1 | grep -v "Warning\|amdgpu.ids"
for n in 1 2 4 8; do echo "== abl$n"; CLOUDCT_LIB=/root/repo/cloud_transformers_amd/lib/libcloudct_abl$n.so timeout 300 python tools/core_bench.py --S=0 --S=1 2>&1 | grep -v "Warning\|amdgpu.ids"; done

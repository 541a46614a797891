"""One zoo head shape, 20 steps (for rocprofv3 --kernel-trace --stats): zoo_one.py C W dim B N [reduce]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
C, W, dim, B, N = [int(v) for v in sys.argv[1:6]]
reduce = sys.argv[6] if len(sys.argv) > 6 else "max"
H = 16
torch.manual_seed(0)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
st = SplatSliceStep(keys, feat, cot, W, H, dim, reduce)
print(st.launch_tags())
for _ in range(20):
    st.run()
torch.cuda.synchronize()

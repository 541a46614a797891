"""time the passes of ONE zoo shape: zoo_one_pass.py C W dim B N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes
C, W, dim, B, N = [int(v) for v in sys.argv[1:6]]
H = 16
torch.manual_seed(0)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
st = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets="--no-tickets" not in sys.argv)
for _ in range(50):
    st.run()
torch.cuda.synchronize()
p = time_passes(st, iters=50)
print(os.path.basename(_lib.LIB_PATH), sys.argv[1:6], {k: round(v * 1e3, 1) for k, v in p.items()})

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes
B, N, H, C, W, dim = 8, 4096, 64, 16, 32, 2
seed, warm = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(seed)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
for _ in range(warm):
    st.run()
torch.cuda.synchronize()
p = time_passes(st, iters=200)
print("seed", seed, "warm", warm, {k: round(v * 1e3, 1) for k, v in p.items()})
p = time_passes(st, iters=200)
print("again", {k: round(v * 1e3, 1) for k, v in p.items()})

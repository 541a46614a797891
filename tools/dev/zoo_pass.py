"""time one pass of the zoo shapes: zoo_pass.py <pass> (B N pairs fixed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes_back_to_back as time_passes
which = sys.argv[1]
SHAPES = [(4, 128, 2), (4, 32, 3), (16, 64, 2), (16, 16, 3), (16, 16, 2), (32, 8, 3)]
out = []
for B, N in [(8, 4096), (2, 16384)]:
    for C, W, dim in SHAPES:
        torch.manual_seed(0)
        H = 16
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
        st.run(); torch.cuda.synchronize()
        p = time_passes(st, iters=20)
        out.append("%.1f" % (p[which] * 1e3))
print(which, os.environ.get("CT_EXP_WANT", "-"), " ".join(out))

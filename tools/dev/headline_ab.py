"""the headline step (B8 N4096 H64 C16 32^2 max) with and without the arrival tickets, alternating, one box"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes
B, N, H, C, W, dim = 8, 4096, 64, 16, 32, 2
torch.manual_seed(0)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
steps = {tk: SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=tk) for tk in (True, False)}
for rep in range(3):
    for tk in (True, False):
        st = steps[tk]
        for _ in range(200):
            st.run()
        torch.cuda.synchronize()
        p = time_passes(st, iters=200)
        print("tickets" if tk else "plain  ", {k: round(v * 1e3, 1) for k, v in p.items()}, round(sum(p.values()) * 1e3, 1), st.launch_tags())

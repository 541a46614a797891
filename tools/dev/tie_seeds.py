"""Splat(max) backward time of one zoo shape over several random workloads: an exact tie somewhere makes a plane's workgroup redo it
(python tools/dev/tie_seeds.py C W dim B N [H])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes
C, W, dim, B, N = [int(v) for v in sys.argv[1:6]]
H = int(sys.argv[6]) if len(sys.argv) > 6 else 16
res = []
for seed in range(10):
    torch.manual_seed(seed)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
    for _ in range(20):
        st.run()
    torch.cuda.synchronize()
    p = time_passes(st, iters=50)
    res.append(round(p["splat_bwd"] * 1e3, 1))
print(sys.argv[1:], st.launch_tags()["splat_bwd"], "splat_bwd us over 10 seeds:", res)

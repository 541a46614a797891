"""Runs ct_pw_gemm a few times per arrangement on one shape (for tools/kprof.sh): python tools/dev/pw_run.py [B Co Ci N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import ops

B, Co, Ci, N = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 848, 512, 4096)
torch.manual_seed(0)
W = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
x = torch.randn(B, Ci, N, device="cuda")
gy = torch.randn(B, Co, N, device="cuda")
am_w, am_x, am_g = ops.amax(W), ops.amax(x), ops.amax(gy)
for _ in range(6):
    ops.pw_gemm(0, W, x, am_w, am_x, B, Co, Ci, N)
    ops.pw_gemm(1, W, gy, am_w, am_g, B, Co, Ci, N)
    ops.pw_gemm(2, gy, x, am_g, am_x, B, Co, Ci, N)
torch.cuda.synchronize()

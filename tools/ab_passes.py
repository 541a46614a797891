"""A/B helper: time the four passes of the headline step with a given libcloudct build
(CLOUDCT_LIB=path python tools/ab_passes.py [reduce])."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes
reduce = sys.argv[1] if len(sys.argv) > 1 else "max"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(1234)
B, N, H, W, dim = 8, 4096, 64, 32, 2
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
step = SplatSliceStep(keys, feat, cot, W, H, dim, reduce)
step.run(); torch.cuda.synchronize()
p = time_passes(step, iters=50)
print(os.path.basename(os.environ.get("CLOUDCT_LIB", "default")), reduce, {k: round(v * 1e3, 1) for k, v in p.items()}, "sum_us", round(sum(p.values()) * 1e3, 1))

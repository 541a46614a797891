"""One EMD forward (B2 n=16384, eps 0.005, 50 iterations) for rocprofv3 --kernel-trace: per-iteration kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.emd import emdModule
B, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 16384)
torch.manual_seed(0)
a = torch.rand(B, n, 3, device="cuda")
b = torch.rand(B, n, 3, device="cuda")
emd = emdModule()
for _ in range(3):
    d, _ = emd(a, b, 0.005, 50)
torch.cuda.synchronize()

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.layers import multihead_ct as M
torch.manual_seed(0)
CFG = {"1": ([4, 4], [128, 32]), "2": ([16, 16], [64, 16]), "3": ([16, 32], [16, 8])}[sys.argv[1] if len(sys.argv) > 1 else "2"]
m = M.MultiHeadUnion(512, CFG[0], CFG[1], [2, 3], [16, 16]).cuda()
x = torch.randn(8, 512, 4096, device="cuda", requires_grad=True)
pcd = torch.rand(8, 3, 4096, device="cuda") * 2 - 1
ITERS = 100
for _ in range(ITERS):
    m.zero_grad(set_to_none=True); x.grad = None
    out, _ = m(x, pcd); out.square().mean().backward()
torch.cuda.synchronize()

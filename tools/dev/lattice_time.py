"""ct_lattice_fwd / ct_lattice_bwd through the autograd ops at the zoo's block shape (B8 H16 N4096, dim 3): us per call (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import ops
B, H, N, dim = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 16, 4096, 3)
torch.manual_seed(0)
xyz = torch.rand(B, 3, N, device="cuda") * 2 - 1
res = (torch.randn(B, H * 3, N, device="cuda") * 0.3).requires_grad_(True)
log_R = torch.randn(H, 3, device="cuda").requires_grad_(True)
shift = (torch.randn(H, 3, device="cuda") * 0.1).requires_grad_(True)
scales = (1 + 0.2 * torch.randn(H, dim, device="cuda")).requires_grad_(True)
cot = torch.randn(B, H * dim, N, device="cuda")
def fwd():
    return ops.lattice(xyz, res, ops.so3_exp(log_R, 1e-4), shift, scales, None, dim, with_stats=True)
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tf = t(fwd)
keys, lat, st = fwd()
def bwd():
    torch.autograd.grad((lat,), (res, log_R, shift, scales), (cot,), retain_graph=True)
tb = t(bwd)
print("B%d H%d N%d dim%d: forward (so3 + lattice + stats) %.1f us, backward (lattice_bwd + finish + param_sum + so3) %.1f us (eager launches)" % (B, H, N, dim, tf, tb))

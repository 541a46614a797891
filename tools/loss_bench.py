"""Chamfer / EMD timings on the completion and reconstruction shapes:
Chamfer B2 n=m=16384 and B4 8192; EMD same shapes with (eps, iters) = (0.005, 50).
Both are O(n^2) fp32 VALU work (not HBM): reported against the 157.3 TF vector peak with
8 flop per pair for Chamfer (3 sub + 3 fma-or-mul/add + compare/select counted as 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.chamfer import chamfer_with_indices
from cloud_transformers_amd.emd import emdModule


def timeit(f, iters):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B, n in [(2, 16384), (4, 8192)]:
    torch.manual_seed(0)
    a = torch.rand(B, n, 3, device="cuda", requires_grad=True)
    b = torch.rand(B, n, 3, device="cuda")
    t_f = timeit(lambda: chamfer_with_indices(a, b), 20)
    def fb():
        d1, d2, _, _ = chamfer_with_indices(a, b)
        (d1.mean() + d2.mean()).backward()
    t_fb = timeit(fb, 20)
    pairs = 2.0 * B * n * n
    print("chamfer B%d n=%d: fwd %.1f us (%.1f TFLOP/s at 8 flop/pair, %.0f%% of 157.3), fwd+bwd %.1f us"
          % (B, n, t_f * 1e3, pairs * 8 / (t_f * 1e-3) / 1e12, pairs * 8 / (t_f * 1e-3) / 157.3e12 * 100, t_fb * 1e3))
    emd = emdModule()
    t_e = timeit(lambda: emd(a, b, 0.005, 50), 5)
    def eb():
        d, _ = emd(a, b, 0.005, 50)
        d.sqrt().mean().backward()
    t_eb = timeit(eb, 5)
    print("emd     B%d n=%d (eps 0.005, 50 iters): fwd %.2f ms, fwd+bwd %.2f ms" % (B, n, t_e, t_eb))

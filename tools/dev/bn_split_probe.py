"""Training BatchNorm forward / backward at B8 N4096 for the channel counts of the blocks: the one-workgroup-per-channel kernels
(ct_bn_relu_fwd / _bwd) against the split statistics + apply kernels (ct_bn_stats_fwd + ct_bn_apply_fwd, ct_bn_reduce_bwd +
ct_bn_apply_bwd, as the SyncBatchNorm path uses them with world = 1); us per call, HIP events."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, N = 8, 4096
f = ctypes.c_float
for C in (32, 48, 64, 128, 256, 512, 768, 1024):
    x = torch.randn(B, C, N, device="cuda"); y = torch.empty_like(x); gy = torch.randn_like(x); gx = torch.empty_like(x)
    w = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda")
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda"); nbt = torch.zeros(1, device="cuda", dtype=torch.int64)
    mean = torch.empty(C, device="cuda"); rstd = torch.empty(C, device="cuda")
    gw = torch.empty(C, device="cuda"); gb = torch.empty(C, device="cuda")
    loc = torch.empty(2 * C + 1, device="cuda"); cnt = torch.empty(1, device="cuda")
    sg = torch.empty(C, device="cuda"); sgx = torch.empty(C, device="cuda")
    one_f = lambda: _lib.check(lib.ct_bn_relu_fwd(_ptr(x), 0, _ptr(w), _ptr(b), _ptr(rm), _ptr(rv), _ptr(nbt), None, 0, _ptr(y), 0, _ptr(mean), _ptr(rstd),
                                                  B, C, N, f(1e-5), f(0.1), 1, _stream()), "f")
    one_b = lambda: _lib.check(lib.ct_bn_relu_bwd(_ptr(x), 0, _ptr(w), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gy), 0, _ptr(gx), 0, _ptr(gw), _ptr(gb),
                                                  B, C, N, 1, _stream()), "b")
    def split_f():
        _lib.check(lib.ct_bn_stats_fwd(_ptr(x), 0, loc.data_ptr(), loc.data_ptr() + 4 * C, loc.data_ptr() + 8 * C, B, C, N, _stream()), "s")
        _lib.check(lib.ct_bn_apply_fwd(_ptr(x), 0, _ptr(w), _ptr(b), loc.data_ptr(), loc.data_ptr() + 4 * C, loc.data_ptr() + 8 * C, 1, 2 * C + 1,
                                       _ptr(rm), _ptr(rv), _ptr(nbt), None, 0, _ptr(y), 0, _ptr(mean), _ptr(rstd), _ptr(cnt), B, C, N, f(1e-5), f(0.1), 1, _stream()), "a")
    def split_b():
        _lib.check(lib.ct_bn_reduce_bwd(_ptr(x), 0, _ptr(w), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gy), 0, _ptr(sg), _ptr(sgx), B, C, N, 1, _stream()), "r")
        _lib.check(lib.ct_bn_apply_bwd(_ptr(x), 0, _ptr(w), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gy), 0, _ptr(sg), _ptr(sgx), _ptr(cnt), _ptr(gx), 0, B, C, N, 1, _stream()), "ab")
    of, ob = t(one_f), t(one_b)
    y1 = y.clone(); gx1 = gx.clone()
    sf, sb = t(split_f), t(split_b)
    mb = B * C * N * 4 / 1e6
    print("C%4d (%5.1f MB): one kernel fwd %5.1f bwd %5.1f us | split fwd %5.1f bwd %5.1f us | max diff y %.1e gx %.1e" % (
        C, mb, of, ob, sf, sb, float((y - y1).abs().max()), float((gx - gx1).abs().max())), flush=True)

"""Forward / backward-data of four-channel 3D groups: gconv_c4_mfma3_kernel (ct_debug_set_gconv(4): always) against the vector-ALU
kernel (2: never) and float64, times (HIP events) and errors on a set of shapes incl. ragged ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, G, W in [(8, 16, (32, 32, 32)), (2, 16, (32, 32, 32)), (1, 3, (5, 7, 16)), (2, 2, (9, 33, 48)), (3, 1, (1, 1, 16)), (2, 5, (2, 20, 64)), (8, 16, (16, 16, 16))]:
    torch.manual_seed(1)
    x = torch.randn(B, G * 4, *W, device="cuda")
    w = torch.randn(G * 4, 4, 3, 3, 3, device="cuda") * 0.1
    b = torch.randn(G * 4, device="cuda")
    y = torch.empty_like(x); gx = torch.empty_like(x)
    Wa = _lib.int_array(W)
    res = {}
    for flag in (4, 2):
        lib.ct_debug_set_gconv(flag)
        y.fill_(float("nan")); gx.fill_(float("nan"))
        f = t(lambda: _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, 4, 4, 3, Wa, _stream()), "f"))
        d = t(lambda: _lib.check(lib.ct_gconv_bwd_data(_ptr(x), _ptr(w), _ptr(gx), B, G, 4, 4, 3, Wa, _stream()), "d"))
        res[flag] = (f, d, y.clone(), gx.clone())
    lib.ct_debug_set_gconv(0)
    xd = x.double().requires_grad_(True)
    yr = torch.nn.functional.conv3d(xd, w.double(), b.double(), padding=1, groups=G)
    gxr, = torch.autograd.grad(yr, xd, x.double())
    err = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print("B%d G%d %s: mfma fwd %.0f bwd %.0f us | valu fwd %.0f bwd %.0f us | err mfma %.1e %.1e  valu %.1e %.1e" % (
        B, G, "x".join(map(str, W)), res[4][0], res[4][1], res[2][0], res[2][1], err(res[4][2], yr), err(res[4][3], gxr), err(res[2][2], yr), err(res[2][3], gxr)), flush=True)

"""forward / backward-data time of the grouped conv with the K-split kernel taking groups from 16 / 32 / 33 input channels"""
import os, sys
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
SHAPES = [(8, 64, 16, 16, (32, 32)), (8, 16, 16, 16, (16, 16)), (8, 16, 16, 16, (64, 64)), (8, 16, 16, 16, (16, 16, 16)), (8, 16, 32, 32, (8, 8, 8)),
          (2, 16, 32, 32, (8, 8, 8)), (2, 16, 16, 16, (16, 16, 16)), (8, 16, 32, 64, (8, 8, 8)), (8, 16, 32, 64, (8, 8)), (8, 16, 32, 32, (16, 16)), (8, 16, 16, 32, (16, 16))]
for B, G, Ci, Co, W in SHAPES:
    dim = len(W)
    x = torch.randn(B, G * Ci, *W, device="cuda")
    w = torch.randn(G * Co, Ci, *([3] * dim), device="cuda") * 0.05
    b = torch.randn(G * Co, device="cuda")
    y = torch.empty(B, G * Co, *W, device="cuda"); gy = torch.randn_like(y); gx = torch.empty_like(x)
    Wa = _lib.int_array(W)
    row = "B%d G%d %d->%d %s:" % (B, G, Ci, Co, "x".join(map(str, W)))
    outs = []
    for thr in (33, 32, 16):
        lib.ct_debug_set_gconv(thr << 8)
        f = t(lambda: _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, Ci, Co, dim, Wa, _stream()), "f"))
        d = t(lambda: _lib.check(lib.ct_gconv_bwd_data(_ptr(gy), _ptr(w), _ptr(gx), B, G, Ci, Co, dim, Wa, _stream()), "d"))
        outs.append((y.clone(), gx.clone()))
        row += "  thr%d fwd %.0f bwd_data %.0f |" % (thr, f, d)
    lib.ct_debug_set_gconv(0)
    e = max(float((o[0] - outs[0][0]).abs().max()) for o in outs), max(float((o[1] - outs[0][1]).abs().max()) for o in outs)
    print(row, "max diff %.1e %.1e" % e, flush=True)

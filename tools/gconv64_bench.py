"""Per-pass times (us) of the grouped 3^d convolution at > 32 channels per group — the Res2D / Res3D stacks of the
classifier / inpainter encoders (model_zoo/scanobject/classifier.py:74-92) — through the C ABI, against PyTorch/MIOpen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream

lib = _lib.load()


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [(8, 16, 32, 64, (8, 8, 8)), (8, 16, 32, 32, (8, 8, 8)), (8, 16, 48, 40, (8, 8, 8)), (3, 5, 64, 64, (4, 8, 8)), (8, 16, 64, 64, (16, 16)), (8, 16, 64, 64, (8, 8, 8)), (8, 16, 64, 64, (4, 4, 4)), (8, 16, 64, 64, (2, 2, 2)),
          (8, 16, 32, 64, (8, 8)), (8, 16, 64, 64, (8, 8)), (8, 16, 64, 64, (4, 4)), (2, 16, 64, 64, (8, 8, 8))]
for B, G, Ci, Co, W in SHAPES:
    dim = len(W)
    x = torch.randn(B, G * Ci, *W, device="cuda", requires_grad=True)
    w = (torch.randn(G * Co, Ci, *([3] * dim), device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(G * Co, device="cuda", requires_grad=True)
    y = torch.empty(B, G * Co, *W, device="cuda")
    gy = torch.randn_like(y)
    gx, gw, gb = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    Wa = _lib.int_array(W)
    ok = lib.ct_gconv_supported(B, G, Ci, Co, dim, Wa)
    res = {}
    if ok:
        nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, Ci, Co, dim, Wa)
        ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
        res["fwd"] = t(lambda: _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, Ci, Co, dim, Wa, _stream()), "f"))
        res["bwd_data"] = t(lambda: _lib.check(lib.ct_gconv_bwd_data(_ptr(gy), _ptr(w), _ptr(gx), B, G, Ci, Co, dim, Wa, _stream()), "d"))
        wrw = lambda: _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), nws, B, G, Ci, Co, dim, Wa, _stream()), "w")
        res["bwd_weight"] = t(wrw)
        lib.ct_debug_set_gconv(1)                       # the vector-ALU small-volume kernel / the ring kernel instead of the MFMA one
        res["bwd_weight_valu"] = t(wrw)
        lib.ct_debug_set_gconv(0)
        wrw()
    fn = torch.nn.functional.conv3d if dim == 3 else torch.nn.functional.conv2d
    ref = {"fwd": t(lambda: fn(x, w, b, padding=1, groups=G))}
    yy = fn(x, w, b, padding=1, groups=G)
    ref["fwd+bwd"] = t(lambda: torch.autograd.grad(fn(x, w, b, padding=1, groups=G), (x, w, b), gy))
    if ok:
        _, gw_ref, gb_ref = torch.autograd.grad(fn(x.double(), w.double(), b.double(), padding=1, groups=G), (x, w, b), gy.double())
        res["err_gw"] = float((gw - gw_ref).abs().max() / gw_ref.abs().max())
        res["err_gb"] = float((gb - gb_ref).abs().max() / gb_ref.abs().max())
    flops = 2.0 * B * G * Ci * Co * (3 ** dim) * float(torch.tensor(W).prod())
    print("B%d G%d %d->%d %s | own %s sum %.0f | miopen fwd %.0f fwd+bwd %.0f | fwd GFLOP %.2f" % (
        B, G, Ci, Co, "x".join(map(str, W)), {k: (round(v) if v > 1 else float("%.1e" % v)) for k, v in res.items()}, sum(res[k] for k in ("fwd", "bwd_data", "bwd_weight") if k in res), ref["fwd"], ref["fwd+bwd"], flops / 1e9), flush=True)

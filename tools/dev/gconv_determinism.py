"""Bitwise repeatability of the three grouped-conv passes on every shape the three models run (tools/gconv_model_shapes.py logs
them): each pass three times on the same inputs, outputs compared with torch.equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
SHAPES = [(8, 16, 4, 4, (32, 32, 32)), (8, 16, 16, 16, (16, 16, 16)), (8, 16, 4, 4, (128, 128)), (8, 16, 16, 16, (64, 64)), (8, 16, 64, 64, (8, 8, 8)),
          (8, 16, 64, 64, (4, 4, 4)), (8, 16, 32, 64, (8, 8, 8)), (8, 16, 64, 64, (4, 4)), (8, 16, 64, 64, (2, 2, 2)), (8, 16, 64, 64, (8, 8)),
          (8, 16, 32, 32, (16, 16)), (8, 16, 32, 64, (8, 8)), (8, 16, 16, 32, (16, 16)), (8, 16, 32, 32, (8, 8, 8)), (8, 16, 16, 16, (16, 16))]
SHAPES += [(2,) + s[1:] for s in SHAPES]
bad = 0
for B, G, Ci, Co, W in SHAPES:
    dim = len(W)
    torch.manual_seed(0)
    x = torch.randn(B, G * Ci, *W, device="cuda"); w = torch.randn(G * Co, Ci, *([3] * dim), device="cuda") * 0.05
    b = torch.randn(G * Co, device="cuda"); gy = torch.randn(B, G * Co, *W, device="cuda")
    Wa = _lib.int_array(W)
    nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, Ci, Co, dim, Wa)
    res = []
    for rep in range(3):
        y = torch.full((B, G * Co) + W, float("nan"), device="cuda"); gx = torch.full_like(x, float("nan"))
        gw = torch.full_like(w, float("nan")); gb = torch.full_like(b, float("nan"))
        ws = torch.full((max(nws, 4) // 4,), float("nan"), device="cuda")          # a workspace full of NaNs: nothing may be read before it is written
        _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, Ci, Co, dim, Wa, _stream()), "f")
        _lib.check(lib.ct_gconv_bwd_data(_ptr(gy), _ptr(w), _ptr(gx), B, G, Ci, Co, dim, Wa, _stream()), "d")
        _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), nws, B, G, Ci, Co, dim, Wa, _stream()), "w")
        torch.cuda.synchronize()
        res.append((y, gx, gw, gb))
    ok = [all(torch.equal(res[0][k], r[k]) for r in res[1:]) for k in range(4)]
    fin = [bool(torch.isfinite(res[0][k]).all()) for k in range(4)]
    if not (all(ok) and all(fin)):
        bad += 1
    print("B%d G%d %d->%d %s: repeatable fwd/bwd_data/bwd_weight/bias %s  finite %s" % (B, G, Ci, Co, "x".join(map(str, W)), ok, fin), flush=True)
print("shapes with a difference:", bad)

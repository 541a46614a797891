"""Weight gradient of four-channel 3D groups: gconv_c4_wrw_mfma3_kernel against the ring kernel's vector-ALU engine
(ct_debug_set_gconv(2)) and float64: times and errors."""
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
for B, G, W in [(8, 16, (32, 32, 32)), (2, 16, (32, 32, 32)), (1, 3, (5, 7, 16)), (2, 2, (9, 33, 48)), (3, 1, (1, 1, 16)), (2, 5, (2, 20, 64)), (8, 16, (16, 16, 16)), (1, 2, (3, 6, 4))]:
    torch.manual_seed(1)
    x = torch.randn(B, G * 4, *W, device="cuda")
    gy = torch.randn(B, G * 4, *W, device="cuda")
    Wa = _lib.int_array(W)
    nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, 4, 4, 3, Wa)
    ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
    res = {}
    for flag in (4, 2):
        lib.ct_debug_set_gconv(flag)
        gw = torch.full((G * 4, 4, 3, 3, 3), float("nan"), device="cuda"); gb = torch.full((G * 4,), float("nan"), device="cuda")
        f = lambda: _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), nws, B, G, 4, 4, 3, Wa, _stream()), "w")
        us = t(f)
        res[flag] = (us, gw.clone(), gb.clone())
    lib.ct_debug_set_gconv(0)
    wd = torch.zeros(G * 4, 4, 3, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    bd = torch.zeros(G * 4, dtype=torch.float64, device="cuda", requires_grad=True)
    yr = torch.nn.functional.conv3d(x.double(), wd, bd, padding=1, groups=G)
    gwr, gbr = torch.autograd.grad(yr, (wd, bd), gy.double())
    err = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print("B%d G%d %s: mfma %.0f us | valu %.0f us | err mfma gw %.1e gb %.1e  valu gw %.1e gb %.1e" % (
        B, G, "x".join(map(str, W)), res[4][0], res[2][0], err(res[4][1], gwr), err(res[4][2], gbr), err(res[2][1], gwr), err(res[2][2], gbr)), flush=True)

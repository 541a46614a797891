"""A few launches of both C4 32^3 weight-gradient kernels for tools/kprof.sh (bash tools/kprof.sh c4w tools/dev/c4_wrw_run.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
B, G, W = 8, 16, (32, 32, 32)
x = torch.randn(B, G * 4, *W, device="cuda"); gy = torch.randn(B, G * 4, *W, device="cuda")
gw = torch.empty(G * 4, 4, 3, 3, 3, device="cuda"); gb = torch.empty(G * 4, device="cuda"); Wa = _lib.int_array(W)
nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, 4, 4, 3, Wa); ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
for flag in (4, 2):
    lib.ct_debug_set_gconv(flag)
    for _ in range(12):
        _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), nws, B, G, 4, 4, 3, Wa, _stream()), "w")
torch.cuda.synchronize()

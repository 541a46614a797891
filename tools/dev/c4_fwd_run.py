"""A few launches of both C4 32^3 forward kernels for tools/kprof.sh (bash tools/kprof.sh c4 tools/dev/c4_fwd_run.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
B, G, W = 8, 16, (32, 32, 32)
x = torch.randn(B, G * 4, *W, device="cuda"); w = torch.randn(G * 4, 4, 3, 3, 3, device="cuda") * 0.1; b = torch.randn(G * 4, device="cuda")
y = torch.empty_like(x); Wa = _lib.int_array(W)
for flag in (0, 2):
    lib.ct_debug_set_gconv(flag)
    for _ in range(12):
        _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, 4, 4, 3, Wa, _stream()), "f")
torch.cuda.synchronize()

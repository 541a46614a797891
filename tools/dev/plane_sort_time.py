"""Time ct_plane_sort alone over batch sizes (how many workgroups per CU does it get?)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream

lib = _lib.load()
H, N, dim, W = 64, 4096, 2, 32
Wa = _lib.int_array([W, W])
for B in (1, 2, 4, 8, 16):
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    n = lib.ct_plane_sort_bytes(B, H, N, dim, Wa)
    rec = torch.empty(n, device="cuda", dtype=torch.uint8)
    f = lambda: _lib.check(lib.ct_plane_sort(_ptr(keys), _ptr(rec), n, B, H, N, dim, Wa, _stream()), "sort")
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("B %2d planes %4d: %.1f us" % (B, B * H, e0.elapsed_time(e1) / 50 * 1e3), flush=True)

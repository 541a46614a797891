"""Phase timeline of the LDS-resident backward from a CT_CORE_STAMP build (see tools/dev/core_stamps.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
NAMES = ["P0 zero", "P1 scatter z", "P2 K", "P3 conv", "P4 channel max", "P5 slice bwd", "P6 g_y + g_b", "P7 conv^T", "P8 wgrad", "P9 splat bwd"]
lib = _lib.load()
dim, W, C = 2, 16, 16
Wa = _lib.int_array([W, W])
for B, H, N in [(8, 16, 4096), (8, 16, 2048)]:
    torch.manual_seed(0)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda")); feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda"); w = torch.randn(H * C, C, 3, 3, device="cuda") / 12; bias = torch.randn(H * C, device="cuda") * 0.1
    g_feat, g_keys, g_w, g_b = torch.empty_like(feat), torch.empty_like(keys), torch.empty_like(w), torch.empty_like(bias)
    nf = lib.ct_mhct_core_bwd_fused_workspace_bytes(B, H, C, N, dim, Wa)
    wsf = torch.zeros(nf, device="cuda", dtype=torch.uint8)
    for _ in range(10):
        _lib.check(lib.ct_mhct_core_bwd_fused(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(bias), _ptr(cot), _ptr(g_feat), _ptr(g_keys),
                                              _ptr(g_w), _ptr(g_b), _ptr(wsf), nf, B, H, C, N, dim, Wa, _stream()), "bwdf")
    torch.cuda.synchronize()
    n = B * H
    st = np.frombuffer(wsf.cpu().numpy()[nf - n * 128:].tobytes(), dtype=np.uint64).reshape(n, 16)
    d = np.diff(st[:, :11].astype(np.int64), axis=1) * 0.01
    print(f"B{B} H{H} N{N}: span {(st[:, 10].max() - st[:, 0].min()) * 0.01:.1f} us")
    for k in range(10):
        print(f"    {NAMES[k]:16s} median {np.median(d[:, k]):6.1f} us")

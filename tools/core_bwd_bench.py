"""Backward of the MHCT core on the 16^2 C16 planes: the LDS-resident kernel (ct_mhct_core_bwd_fused: recomputes z and conv(z),
nothing saved by the forward) against the backward from saved grids (ct_mhct_core_bwd = Slice backward -> conv backward data /
weight -> Splat backward on this library's kernels); HIP events, us per call.  Also the forward + backward pairs: fused forward
without side outputs + LDS-resident backward vs fused forward writing z, y + backward from saved grids vs the unfused chain."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
from core_bench import timeit


def main():
    lib = _lib.load()
    dim, W, C = 2, 16, 16
    Wa = _lib.int_array([W, W])
    for B, H, N in [(8, 16, 4096), (8, 16, 2048), (8, 64, 4096), (2, 16, 16384)]:
        torch.manual_seed(0)
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        w = torch.randn(H * C, C, 3, 3, device="cuda") / 12
        bias = torch.randn(H * C, device="cuda") * 0.1
        out, g_feat, g_keys = torch.empty_like(feat), torch.empty_like(feat), torch.empty_like(keys)
        g_w, g_b = torch.empty_like(w), torch.empty_like(bias)
        z = torch.empty(B, H * C, W, W, device="cuda"); y = torch.empty_like(z)
        occ = torch.empty((), device="cuda", dtype=torch.int64)
        nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
        ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
        _lib.check(lib.ct_mhct_core_workspace_init(_ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "init")
        nb = lib.ct_mhct_core_bwd_workspace_bytes(B, H, C, N, dim, Wa)
        wsb = torch.empty(nb, device="cuda", dtype=torch.uint8)
        nf = lib.ct_mhct_core_bwd_fused_workspace_bytes(B, H, C, N, dim, Wa)
        wsf = torch.empty(nf, device="cuda", dtype=torch.uint8)

        def fwd(save):
            _lib.check(lib.ct_mhct_core_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(bias), _ptr(out), _ptr(z) if save else None,
                                            _ptr(y) if save else None, _ptr(occ), _ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "fwd")

        def bwd_saved():
            _lib.check(lib.ct_mhct_core_bwd(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(z), _ptr(y), _ptr(cot), _ptr(g_feat), _ptr(g_keys),
                                            _ptr(g_w), _ptr(g_b), _ptr(wsb), nb, B, H, C, N, dim, Wa, _stream()), "bwd")

        def bwd_fused():
            _lib.check(lib.ct_mhct_core_bwd_fused(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(bias), _ptr(cot), _ptr(g_feat), _ptr(g_keys),
                                                  _ptr(g_w), _ptr(g_b), _ptr(wsf), nf, B, H, C, N, dim, Wa, _stream()), "bwdf")

        fwd(True)
        t_s, t_f = timeit(bwd_saved, 100), timeit(bwd_fused, 100)
        p_s = timeit(lambda: (fwd(True), bwd_saved()), 100)
        p_f = timeit(lambda: (fwd(False), bwd_fused()), 100)
        print(f"B{B} H{H} N{N} 16^2 C16: backward from saved grids {t_s:6.1f} us | LDS-resident {t_f:6.1f} us ({t_s / t_f:4.2f}x) || "
              f"fwd+bwd saved {p_s:6.1f} us | recompute {p_f:6.1f} us ({p_s / p_f:4.2f}x)", flush=True)


if __name__ == "__main__":
    main()

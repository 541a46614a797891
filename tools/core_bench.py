"""A/B of the plane-resident MHCT core (ct_mhct_core_fwd) against this package's unfused chain
(ct_splat_fwd -> ct_grid_occupancy -> ct_gconv_fwd -> ct_slice_fwd) on the shapes it is built for: HIP-event time per
forward, launches on torch's current stream, inputs resident in HBM.  `train` = the fused kernel also writes z and
conv(z) out for the backward; `infer` = they never leave the chip.

    python3 tools/core_bench.py [--profile]        (--profile: few iterations, for rocprofv3 --kernel-trace --stats)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd import _lib, ops
from cloud_transformers_amd.ops import _ptr, _stream

SHAPES = [  # (B, H, N, dim, W, C)
    (8, 64, 4096, 2, 32, 16),
    (8, 16, 4096, 2, 16, 16),
    (8, 16, 4096, 3, 8, 32),
    (8, 16, 2048, 2, 16, 16),
    (8, 16, 2048, 3, 8, 32),
    (2, 16, 16384, 2, 16, 16),
    (2, 16, 16384, 3, 8, 32),
]


def timeit(fn, iters):
    for _ in range(max(3, iters // 10)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    iters = 5 if "--profile" in sys.argv else 200
    lib = _lib.load()
    forced = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--S=")]
    for B, H, N, dim, W, C in SHAPES:
        torch.manual_seed(0)
        Wl = [W] * dim
        Wa = _lib.int_array(Wl)
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        w = torch.randn(H * C, C, *([3] * dim), device="cuda") / (C * 3 ** dim) ** 0.5
        bias = torch.randn(H * C, device="cuda") * 0.1
        out = torch.empty_like(feat)
        z = torch.empty(B, H * C, *Wl, device="cuda")
        y = torch.empty_like(z)
        occ = torch.empty((), device="cuda", dtype=torch.int64)

        def unfused():
            _lib.check(lib.ct_splat_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(z), B, H, C, N, dim, Wa, 0, _stream()), "splat")
            _lib.check(lib.ct_grid_occupancy(_ptr(z), z.numel(), _ptr(occ), _stream()), "occ")
            _lib.check(lib.ct_gconv_fwd(_ptr(z), _ptr(w), _ptr(bias), _ptr(y), B, H, C, C, dim, Wa, _stream()), "gconv")
            _lib.check(lib.ct_slice_fwd(_ptr(keys), _ptr(y), None, 0, _ptr(out), B, H, C, N, dim, Wa, _stream()), "slice")

        t_un = timeit(unfused, iters)
        row = f"B{B} H{H} N{N} {dim}D W{W} C{C}: unfused {t_un:7.1f} us"
        for S in (forced or [0]):
            lib.ct_debug_set_core(S << 8)
            nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
            ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
            _lib.check(lib.ct_mhct_core_workspace_init(_ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "init")

            def fused(train):
                _lib.check(lib.ct_mhct_core_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(bias), _ptr(out),
                                                _ptr(z) if train else None, _ptr(y) if train else None, _ptr(occ), _ptr(ws), nws,
                                                B, H, C, N, dim, Wa, _stream()), "core")

            t_tr = timeit(lambda: fused(True), iters)
            t_in = timeit(lambda: fused(False), iters)
            row += f" | S={S or 'auto'}: fused train {t_tr:7.1f} us ({t_un / t_tr:4.2f}x)  infer {t_in:7.1f} us ({t_un / t_in:4.2f}x)"
        lib.ct_debug_set_core(0)
        print(row, flush=True)


if __name__ == "__main__":
    main()

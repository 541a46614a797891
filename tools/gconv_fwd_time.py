"""Time ct_gconv_fwd alone (µs) on the zoo shapes; CLOUDCT_LIB selects an A/B build."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd import _lib

SHAPES = [("2D 32^2 C16 H64", 8, 64, 16, (32, 32)), ("2D 128^2 C4 H16", 8, 16, 4, (128, 128)), ("2D 64^2 C16 H16", 8, 16, 16, (64, 64)),
          ("2D 16^2 C16 H16", 8, 16, 16, (16, 16)), ("3D 32^3 C4 H16", 8, 16, 4, (32, 32, 32)), ("3D 16^3 C16 H16", 8, 16, 16, (16, 16, 16)),
          ("3D 8^3 C32 H16", 8, 16, 32, (8, 8, 8))]


def main():
    lib = _lib.load()
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, B, G, C, W in SHAPES:
        if only and only not in name:
            continue
        dim = len(W)
        x = torch.randn(B, G * C, *W, device="cuda")
        w = torch.randn(G * C, C, *([3] * dim), device="cuda")
        bias = torch.randn(G * C, device="cuda")
        y = torch.empty_like(x)
        Wa = (ctypes.c_int * dim)(*W)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            _lib.check(lib.ct_gconv_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, G, C, C, dim, Wa, st), "fwd")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        vol = 1
        for v in W:
            vol *= v
        flops = 2.0 * B * vol * G * C * C * 3 ** dim
        byts = 2.0 * B * G * C * vol * 4
        print(f"{name}: fwd {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s ({flops / us / 1e6 / 157.3:.2f} of MFMA peak)  "
              f"{byts / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()

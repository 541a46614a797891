"""Pointwise-convolution GEMMs of the model steps: ct_pw_gemm (split-f16 MFMA terms, fp32 in/out) against the rocBLAS fp32 GEMMs
torch.bmm reaches, per arrangement (forward / data gradient / weight gradient): us per call and the largest error against
the float64 product relative to sum |a b|.  Shapes: the stacked projections of the S3DIS segmenter's three head
configurations at B8 N4096 (tools/segmenter_step_bench.py), the classifier's at N2048 and the decoder's at B2 N16384."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd import ops

SHAPES = [(8, 208, 512, 4096), (8, 592, 512, 4096), (8, 848, 512, 4096), (8, 512, 64, 4096), (8, 512, 256, 4096),
          (8, 512, 512, 4096), (8, 512, 1024, 4096), (8, 512, 128, 4096), (8, 128, 512, 4096), (8, 64, 512, 4096), (8, 848, 512, 2048), (2, 848, 512, 16384), (2, 512, 512, 16384)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]


def timeit(fn, iters=20):
    for _ in range(3):
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
    torch.manual_seed(0)
    print("B Co Ci N | mode | split16 us (TF eff) err | amax us | rocBLAS fp32 us (TF) err")
    for (B, Co, Ci, N) in SHAPES:
        W = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
        x = torch.randn(B, Ci, N, device="cuda")
        gy = torch.randn(B, Co, N, device="cuda")
        am_w, am_x, am_g = ops.amax(W), ops.amax(x), ops.amax(gy)
        flop = 2.0 * B * Co * Ci * N
        Wd, xd, gd = W.double(), x.double(), gy.double()
        refs = {
            0: (torch.matmul(Wd, xd), torch.matmul(Wd.abs(), xd.abs())),
            1: (torch.matmul(Wd.t(), gd), torch.matmul(Wd.abs().t(), gd.abs())),
            2: (torch.matmul(gd, xd.transpose(1, 2)).sum(0), torch.matmul(gd.abs(), xd.abs().transpose(1, 2)).sum(0)),
        }
        del Wd, xd, gd
        mine = {0: lambda: ops.pw_gemm(0, W, x, am_w, am_x, B, Co, Ci, N),
                1: lambda: ops.pw_gemm(1, W, gy, am_w, am_g, B, Co, Ci, N),
                2: lambda: ops.pw_gemm(2, gy, x, am_g, am_x, B, Co, Ci, N)}
        lib = {0: lambda: torch.bmm(W.unsqueeze(0).expand(B, -1, -1), x),
               1: lambda: torch.bmm(W.t().unsqueeze(0).expand(B, -1, -1), gy),
               2: lambda: torch.bmm(gy, x.transpose(1, 2)).sum(0)}
        t_amax = timeit(lambda: ops.amax(x))
        for mode, name in ((0, "fwd  "), (1, "dgrad"), (2, "wgrad")):
            ref, mag = refs[mode]
            e_m = float(((mine[mode]().double() - ref).abs() / (mag + 1e-30)).max())
            e_l = float(((lib[mode]().double() - ref).abs() / (mag + 1e-30)).max())
            tm, tl = timeit(mine[mode]), timeit(lib[mode])
            print("%d %d %d %d | %s | %7.1f (%5.1f) %.1e | %5.1f | %7.1f (%5.1f) %.1e | x%.2f" %
                  (B, Co, Ci, N, name, tm, flop / tm * 1e-6, e_m, t_amax, tl, flop / tl * 1e-6, e_l, tl / tm))
        del refs


if __name__ == "__main__":
    main()

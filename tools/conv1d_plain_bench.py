"""The plain nn.Conv1d(kernel_size=1) layers of the reference's model files (stems, heads: model_zoo/*/*.py) — torch / MIOpen
against this package's PointwiseConv1d (three batched GEMMs) — fwd+bwd, steady state (MIOpen's find-mode runs excluded by the
warm-up), µs per call."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers.pointwise import PointwiseConv1d

SHAPES = [  # (B, Cin, Cout, N, bias)
    (8, 6, 512, 4096, True), (8, 512, 512, 4096, False), (8, 512, 13, 4096, True),                # S3DIS segmenter stem / head
    (8, 3, 512, 2048, False), (8, 1536, 256, 2048, False), (8, 256, 1, 2048, True),               # ScanObjectNN classifier stem / mask head
    (2, 4, 512, 16384, False), (2, 516, 512, 16384, False), (2, 512, 3, 16384, True),             # completion decoder stem / head
    (4, 3, 512, 8192, False), (4, 512, 512, 8192, False),                                           # What3D decoder
]


def timeit(fn, iters=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    for B, ci, co, N, bias in SHAPES:
        x = torch.randn(B, ci, N, device="cuda", requires_grad=True)
        gy = torch.randn(B, co, N, device="cuda")
        res = []
        for cls in (nn.Conv1d, PointwiseConv1d):
            m = cls(ci, co, kernel_size=1, bias=bias).cuda()

            def step():
                x.grad = None
                m.zero_grad(set_to_none=True)
                m(x).backward(gy)
            res.append(timeit(step))
        print(f"B{B} {ci:4d} -> {co:3d} N{N}: nn.Conv1d {res[0]:8.1f} us | PointwiseConv1d {res[1]:8.1f} us  ({res[0] / res[1]:.2f}x)", flush=True)


if __name__ == "__main__":
    main()

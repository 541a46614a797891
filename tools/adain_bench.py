"""AdaIN (+ReLU) forward+backward: the fused kernels vs torch's InstanceNorm1d composition, decoder shapes.
Prints µs and the fraction of the 8 TB/s roofline (algorithmic bytes: 8 B/element forward, 12 backward)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cloud_transformers_amd import ops


def timeit(fn, iters=50):
    for _ in range(5):
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
    inorm = torch.nn.InstanceNorm1d(1, eps=1e-5, affine=False)
    for B, C, N in [(2, 512, 16384), (4, 512, 8192), (2, 256, 16384), (2, 48, 16384), (8, 512, 4096), (2, 512, 2048)]:
        x = torch.randn(B, C, N, device="cuda", requires_grad=True)
        gb = torch.randn(B, 2, C, device="cuda", requires_grad=True)
        gy = torch.randn(B, C, N, device="cuda")

        def fused_fwd():
            return ops.adain(x, gb, 1e-5, True)

        def torch_fwd():
            return torch.relu(inorm(x) * (gb[:, 0, :, None] + 1) + gb[:, 1, :, None])

        def fb(f):
            def run():
                x.grad = None
                gb.grad = None
                f().backward(gy)
            return run
        with torch.no_grad():
            tf, tt = timeit(fused_fwd), timeit(torch_fwd)
        tfb, ttb = timeit(fb(fused_fwd)), timeit(fb(torch_fwd))
        el = B * C * N
        print(f"B{B} C{C} N{N}: fwd {tf:.1f} us ({8 * el / tf / 8e6:.2f} of 8 TB/s) vs torch {tt:.1f} | "
              f"fwd+bwd {tfb:.1f} us ({20 * el / tfb / 8e6:.2f}) vs torch {ttb:.1f} -> x{ttb / tfb:.2f}")


if __name__ == "__main__":
    main()

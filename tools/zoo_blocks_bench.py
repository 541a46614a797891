"""The three MultiHeadUnion block types of the reference's S3DIS / ScanObjectNN model zoo (model_zoo/s3dis/segmenter.py:28-45:
H16 heads, model_dim 512; stage 1 (C4, 128^2) + (C4, 32^3); stage 2 (C16, 64^2) + (C16, 16^3); stage 3 (C16, 16^2) + (C32, 8^3),
four blocks per stage) fwd+bwd at the S3DIS per-GPU batch B8 N4096: ms per block, eager launches and as one HIP graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers import multihead_ct as M

STAGES = [("stage 1: (C4,128^2)+(C4,32^3)", [4, 4], [128, 32]), ("stage 2: (C16,64^2)+(C16,16^3)", [16, 16], [64, 16]),
          ("stage 3: (C16,16^2)+(C32,8^3)", [16, 32], [16, 8])]


def timeit(fn, iters=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.manual_seed(0)
    x = torch.randn(B, 512, N, device="cuda", requires_grad=True)
    pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1
    total_e = total_g = 0.0
    for name, feats, sizes in STAGES:
        m = M.MultiHeadUnion(512, feats, sizes, [2, 3], [16, 16]).cuda()

        def step():
            m.zero_grad(set_to_none=True)
            x.grad = None
            out, _ = m(x, pcd)
            out.square().mean().backward()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(4):
                step()
        torch.cuda.current_stream().wait_stream(s)
        eager = timeit(step)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        g.replay()
        graphed = timeit(g.replay)
        total_e += eager
        total_g += graphed
        print(f"{name}: eager {eager:.2f} ms | one HIP graph {graphed:.2f} ms", flush=True)
        del g, m
    print(f"12-block stack (4 per stage) fwd+bwd B{B} N{N}: eager {4 * total_e:.1f} ms | graphed {4 * total_g:.1f} ms "
          f"({B * N / (4 * total_g) / 1e3:.2f} M points/s)")


if __name__ == "__main__":
    main()

"""Decoder block of the completion / reconstruction models: MultiHeadUnionAdaIn(512, [16,16], [64,16], [2,3], [16,16], n_latent 256)
fwd+bwd at the decoder shapes B2 N16384 (inpainter) and B4 N8192 (What3D): ms per block, eager launches and as one HIP graph.
`python tools/adain_block_bench.py prof` runs 60 plain iterations for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers import multihead_ct as M


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
    prof = len(sys.argv) > 1 and sys.argv[1] == "prof"
    for B, N in ([(2, 16384)] if prof else [(2, 16384), (4, 8192)]):
        torch.manual_seed(0)
        m = M.MultiHeadUnionAdaIn(512, [16, 16], [64, 16], [2, 3], [16, 16], n_latent=256).cuda()
        x = torch.randn(B, 512, N, device="cuda", requires_grad=True)
        style = torch.randn(B, 256, device="cuda")
        pcd = torch.nn.functional.normalize(torch.randn(B, 3, N, device="cuda"), dim=1)     # sphere noise

        def step():
            m.zero_grad(set_to_none=True)
            x.grad = None
            out, _ = m(x, style, pcd)
            out.square().mean().backward()
        if prof:
            for _ in range(60):
                step()
            torch.cuda.synchronize()
            return
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
        print(f"MultiHeadUnionAdaIn fwd+bwd B{B} N{N}: eager {eager:.2f} ms | one HIP graph {graphed:.2f} ms ({B * N / graphed:.0f} k points/s)",
              flush=True)


if __name__ == "__main__":
    main()

"""MultiHeadUnion fwd+bwd replayed as ONE HIP graph (torch.cuda.CUDAGraph): every libcloudct launch goes to
torch's current stream, so the whole step — rocBLAS / MIOpen nodes and ours — captures and replays without
host launch overhead.  Prints eager vs graphed ms per step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers import multihead_ct as M


def main():
    small = len(sys.argv) > 1 and sys.argv[1] == "small"          # tests/test_graph_gpu.py: a quick functional run
    B, N, dim = (2, 512, 64) if small else (8, 4096, 512)
    torch.manual_seed(1)
    m = (M.MultiHeadUnion(dim, [4, 8], [16, 8], [2, 3], [4, 4]) if small else M.MultiHeadUnion(dim, [16, 16], [64, 16], [2, 3], [16, 16])).cuda()
    x = torch.randn(B, dim, N, device="cuda", requires_grad=True)
    pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1

    def step():
        m.zero_grad(set_to_none=True)
        x.grad = None
        out, _ = m(x, pcd)
        out.square().mean().backward()

    def timeit(fn, iters=20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):          # warm-up on the side stream: MIOpen / rocBLAS pick their kernels, allocator settles
            step()
    torch.cuda.current_stream().wait_stream(s)
    eager = timeit(step)
    m.eval()                        # BatchNorm on its running statistics: the step is a pure function of (x, pcd, weights)
    step()
    ref = [p.grad.clone() for p in m.parameters() if p.grad is not None]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):       # capture records, it does not run
        step()
    g.replay()
    torch.cuda.synchronize()
    now = [p.grad for p in m.parameters() if p.grad is not None]
    same = len(ref) == len(now) and all(torch.allclose(a, b, rtol=1e-3, atol=1e-6) for a, b in zip(ref, now))
    m.train()                       # timing in training mode, like the eager number
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        step()
    g2.replay()
    graphed = timeit(g2.replay)
    print(f"MultiHeadUnion fwd+bwd B{B} N{N}: eager launches {eager:.2f} ms | one HIP graph {graphed:.2f} ms "
          f"({B * N / graphed:.0f} k points/s) | replay reproduces the gradients: {same}")


if __name__ == "__main__":
    main()

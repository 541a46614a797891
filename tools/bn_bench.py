"""Training-mode BatchNorm1d + ReLU, forward + backward as one HIP graph: the fused ct_bn_relu kernels against torch's
BatchNorm1d (MIOpen) + ReLU at the blocks' shapes — register-resident channels (B*N <= 32768) and the loop kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd import ops
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for B, C, N in [(8, 512, 4096), (16, 512, 4096), (32, 512, 2048), (4, 512, 16384)]:
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    relu = torch.nn.ReLU()
    x = torch.randn(B, C, N, device="cuda", requires_grad=True)
    g = torch.randn(B, C, N, device="cuda")
    def fused():
        x.grad = None
        ops.bn_relu(x, bn, True).backward(g)
    def lib():
        x.grad = None
        relu(bn(x)).backward(g)
    res = []
    for fn in (fused, lib):
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s_)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        res.append(t(gr.replay))
    print(f"B{B} C{C} N{N}: fused fwd+bwd {res[0]:.0f} us | torch BatchNorm1d + ReLU {res[1]:.0f} us (graphed)")

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes
B, N, H, C, W, dim = 8, 4096, 64, 16, 32, 2
torch.manual_seed(0)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
step = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
step.run(); torch.cuda.synchronize()
graphs = {}
for gs in (1, 10, 20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(gs):
            step.run()
    g.replay(); torch.cuda.synchronize()
    graphs[gs] = g
time_passes(step)
def trial(gs, K=20):
    g = graphs[gs]
    for _ in range(max(1, 5 // gs)):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K // gs):
        g.replay()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e3, e0.elapsed_time(e1), (t1 - t0) * 1e3
for gs in (1, 10, 20, 10, 20, 1):
    r = [trial(gs) for _ in range(5)]
    r.sort()
    host, dev, enq = r[2]
    print("gs %2d: host %.3f ms  device(events) %.3f ms  enqueue returned after %.3f ms | per step host %.4f dev %.4f" % (gs, host, dev, enq, host / 20, dev / 20))
# eager
def eager(K=20):
    for _ in range(5): step.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): step.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e3, (t1 - t0) * 1e3
r = sorted(eager() for _ in range(5))
print("eager: host %.3f ms, enqueue %.3f ms, per step %.4f" % (r[2][0], r[2][1], r[2][0] / 20))

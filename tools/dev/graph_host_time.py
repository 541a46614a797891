"""tools/dev/graph_host_time.py: is a replayed training step fed fast enough by the host?  The segmenter step as ONE HIP graph: host time
spent inside graph.replay() (the runtime enqueues the graph's nodes), device time per step, and the same with the host running ahead
(several replays enqueued before the first synchronisation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
from torch import nn
from segmenter_step_bench import Segmenter
from cloud_transformers_amd.layers.pointwise import convert_pointwise
B, N = 8, 4096
torch.manual_seed(0)
net = convert_pointwise(Segmenter().cuda())
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
cloud = torch.cat([torch.rand(B, 3, N, device="cuda") * 2 - 1, torch.rand(B, 3, N, device="cuda")], dim=1)
labels = torch.randint(13, (B, N), device="cuda")
lossf = nn.CrossEntropyLoss()
def fwd_bwd():
    opt.zero_grad(set_to_none=True)
    loss = lossf(net(cloud), labels)
    loss.backward()
    return loss
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fwd_bwd(); opt.step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd_bwd()
g.replay(); torch.cuda.synchronize()
# host time inside replay(), one replay at a time
host = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dev = []
for _ in range(8):
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    host.append((t1 - t0) * 1e3); dev.append(e0.elapsed_time(e1))
print("one replay at a time: host time in graph.replay() %.2f ms (min %.2f), device time of the replay %.2f ms (min %.2f)" % (
    sum(host) / len(host), min(host), sum(dev) / len(dev), min(dev)))
# the host running ahead: 8 replays enqueued back to back
torch.cuda.synchronize()
e0.record(); t0 = time.perf_counter()
for _ in range(8):
    g.replay()
t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
print("8 replays back to back: host %.2f ms per replay, device %.2f ms per replay" % ((t1 - t0) * 1e3 / 8, e0.elapsed_time(e1) / 8))

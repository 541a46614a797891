"""A full training step of the ScanObjectNN classifier at BASELINE config 2's shapes (configs/scanobjectnn.yaml: 2048-point
clouds): stem, twelve MultiHeadUnion blocks, the 2D / 3D MultiHeadPool heads with their grouped Res stacks, class head and
per-point mask head; loss = cross-entropy on the 15 classes + binary cross-entropy on the foreground mask
(train_classification.py's two terms), backward, SGD step.  The network is the tests' restatement of
model_zoo/scanobject/classifier.py (tests/test_zoo_gpu.py::Classifier, pinned on the reference's outputs); synthetic clouds,
random-initialised weights.  Prints ms per step with eager launches and with forward + loss + backward as one HIP graph."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers.pointwise import convert_pointwise          # noqa: E402
from tests.test_zoo_gpu import Classifier          # noqa: E402


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 2048
    torch.manual_seed(0)
    net = convert_pointwise(Classifier().cuda()).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    cloud = torch.rand(B, 3, 1, N, device="cuda") * 2 - 1
    labels = torch.randint(15, (B,), device="cuda")
    fg = (torch.rand(B, 1, 1, N, device="cuda") > 0.4).float()
    ce, bce = nn.CrossEntropyLoss(), nn.BCEWithLogitsLoss()
    nparam = sum(p.numel() for p in net.parameters())

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        logits, mask = net(cloud)
        loss = ce(logits, labels) + bce(mask, fg)
        loss.backward()
        return loss

    def step():
        fwd_bwd()
        opt.step()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    eager = timeit(step, 5)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
    g.replay()

    def graphed_step():
        g.replay()
        opt.step()
    graphed = timeit(graphed_step, 5)
    print(f"ScanObjectNN classifier ({nparam / 1e6:.1f} M parameters, 12 MultiHeadUnion blocks + 2D/3D pooling heads), B{B} N{N}, "
          f"1x MI355X, fp32: training step eager {eager:.1f} ms ({B * N / eager:.0f} k points/s) | fwd+loss+bwd as one HIP graph + "
          f"optimizer {graphed:.1f} ms ({B * N / graphed:.0f} k points/s, {B / graphed * 1e3:.0f} clouds/s) | loss {float(static_loss.detach()):.3f}")


if __name__ == "__main__":
    main()

"""A full training step of an S3DIS-segmenter-shaped network built from this package's blocks (BASELINE config 3's
per-GPU work: 4096-point clouds, batch 8): stem Conv1d(6->512)+BN+ReLU, twelve MultiHeadUnion blocks cycling the
zoo's three head configurations (model_zoo/s3dis/segmenter.py:28-45), Conv-BN-ReLU-Conv head to 13 classes;
cross-entropy loss, backward, SGD step; synthetic data, random-initialised weights.  Prints ms per step and points/s
on one MI355X: eager launches, and forward+backward replayed as one HIP graph with the optimizer step outside it."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers.multihead_ct import MultiHeadUnion

ZOO = [([4, 4], [128, 32]), ([16, 16], [64, 16]), ([16, 32], [16, 8])]


class Segmenter(nn.Module):
    def __init__(self, n_classes=13, dim=512, repeats=4):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv1d(6, dim, 1), nn.BatchNorm1d(dim), nn.ReLU(inplace=True))
        self.blocks = nn.ModuleList([MultiHeadUnion(dim, f, s, [2, 3], [16, 16], model_dim_out=dim)
                                     for _ in range(repeats) for f, s in ZOO])
        self.head = nn.Sequential(nn.Conv1d(dim, dim, 1, bias=False), nn.BatchNorm1d(dim), nn.ReLU(inplace=True),
                                  nn.Conv1d(dim, n_classes, 1))

    def forward(self, cloud):                      # cloud [B, 6, N]: xyz + rgb
        x = self.stem(cloud)
        xyz = cloud[:, :3].contiguous()
        for blk in self.blocks:
            x, _ = blk(x, xyz)
        return self.head(x)


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
    B, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.manual_seed(0)
    net = Segmenter().cuda()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    cloud = torch.cat([torch.rand(B, 3, N, device="cuda") * 2 - 1, torch.rand(B, 3, N, device="cuda")], dim=1)
    labels = torch.randint(13, (B, N), device="cuda")
    lossf = nn.CrossEntropyLoss()
    nparam = sum(p.numel() for p in net.parameters())

    def fwd_bwd():
        opt.zero_grad(set_to_none=False)
        loss = lossf(net(cloud), labels)
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
    print(f"S3DIS-shaped segmenter ({nparam / 1e6:.1f} M parameters, 12 MultiHeadUnion blocks), B{B} N{N}, 1x MI355X, fp32: "
          f"training step eager {eager:.1f} ms ({B * N / eager:.0f} k points/s) | fwd+bwd as one HIP graph + optimizer {graphed:.1f} ms "
          f"({B * N / graphed:.0f} k points/s) | loss {float(static_loss):.3f}")


if __name__ == "__main__":
    main()

"""A full training step of an S3DIS-segmenter-shaped network built from this package's blocks (BASELINE config 3's
per-GPU work: 4096-point clouds, batch 8): stem Conv1d(6->512)+BN+ReLU, twelve MultiHeadUnion blocks cycling the
zoo's three head configurations (model_zoo/s3dis/segmenter.py:28-45), Conv-BN-ReLU-Conv head to 13 classes;
cross-entropy loss, backward, SGD step; synthetic data, random-initialised weights.  Prints ms per step and points/s
on one MI355X: eager launches, and forward+backward replayed as one HIP graph with the optimizer step outside it.

Data parallel (the reference's recipe, train_segmentation.py:58-61,128-130): launched by
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/segmenter_step_bench.py`
every rank builds the same network, converts it to SyncBatchNorm, wraps it in DistributedDataParallel (RCCL) and steps its
own batch of 8 clouds; rank 0 prints the max-over-ranks step time and the whole-job points/s (weak scaling)."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers.pointwise import convert_pointwise
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


def main_ddp(world, rank, local_rank, N):
    import torch.distributed as dist
    from cloud_transformers_amd.parallel import barrier, data_parallel, max_over_ranks
    B = 8
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.manual_seed(0)                               # same initial weights on every rank
    net = convert_pointwise(Segmenter().cuda())
    ddp = data_parallel(net, local_rank, broadcast_buffers=os.environ.get("CT_DDP_BCAST", "0") == "1")
    opt = torch.optim.SGD(ddp.parameters(), lr=0.01, momentum=0.9)
    torch.manual_seed(1234 + rank)                     # its own shard of the batch
    cloud = torch.cat([torch.rand(B, 3, N, device="cuda") * 2 - 1, torch.rand(B, 3, N, device="cuda")], dim=1)
    labels = torch.randint(13, (B, N), device="cuda")
    lossf = nn.CrossEntropyLoss()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = lossf(ddp(cloud), labels)
        loss.backward()                                # bucketed gradient all-reduce overlaps with this
        opt.step()
        return loss

    for _ in range(3):
        step()
    steps = 10
    barrier(dist)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    barrier(dist)
    dt = max_over_ranks(dist, time.perf_counter() - t0) / steps
    nbytes = sum(p.numel() for p in net.parameters()) * 4
    if rank == 0:
        print(f"S3DIS-shaped segmenter, DDP + SyncBatchNorm over RCCL, {world} x MI355X, B{B} N{N} per GPU, fp32: "
              f"{dt * 1e3:.1f} ms per step, {world * B * N / dt / 1e3:.0f} k points/s whole job "
              f"(gradient all-reduce {nbytes / 1e6:.1f} MB per step) | loss {float(loss):.3f}")
    dist.barrier()
    dist.destroy_process_group()


def main():
    B, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return main_ddp(int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), N)
    torch.manual_seed(0)
    net = convert_pointwise(Segmenter().cuda())
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    cloud = torch.cat([torch.rand(B, 3, N, device="cuda") * 2 - 1, torch.rand(B, 3, N, device="cuda")], dim=1)
    labels = torch.randint(13, (B, N), device="cuda")
    lossf = nn.CrossEntropyLoss()
    nparam = sum(p.numel() for p in net.parameters())

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)       # fresh gradients each step (the graph re-creates them in place): no zero fills, no accumulation adds
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
    # the WHOLE step — forward, loss, backward AND the optimizer — as one graph: no eager launches between replays (the hand-off
    # graph -> eager kernels -> next graph idles the GPU 0.4-0.9 ms per step: tools/dev/step_timeline.py).  Valid where the learning
    # rate is a constant or a device tensor (a Python-float rate is baked into the captured kernels).
    whole = None
    if os.environ.get("CT_STEP_WHOLE_GRAPH", "1") != "0":
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            fwd_bwd()
            opt.step()
        g2.replay()
        whole = timeit(g2.replay, 5)
    print(f"S3DIS-shaped segmenter ({nparam / 1e6:.1f} M parameters, 12 MultiHeadUnion blocks), B{B} N{N}, 1x MI355X, fp32: "
          f"training step eager {eager:.1f} ms ({B * N / eager:.0f} k points/s) | fwd+bwd as one HIP graph + optimizer {graphed:.1f} ms "
          f"({B * N / graphed:.0f} k points/s) | loss {float(static_loss):.3f}"
          + (f" | whole step (optimizer inside) as one HIP graph {whole:.2f} ms ({B * N / whole:.0f} k points/s)" if whole else ""))


if __name__ == "__main__":
    main()

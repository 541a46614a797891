"""One rank of tests/test_bench_ddp_cpu.py: bench.ddp_step_measure on a CPU stand-in of the segmenter under gloo (started by
cloud_transformers_amd.launch.spawn_ranks, which sets RANK / WORLD_SIZE / MASTER_*); rank 0 writes the bench line's `ddp_step`
object to argv[1]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from torch import nn

import bench


def make(rank):
    net = nn.Sequential(nn.Conv1d(6, 8, 1), nn.GroupNorm(2, 8), nn.ReLU(), nn.Conv1d(8, 13, 1))      # (SyncBatchNorm is GPU-only)
    g = torch.Generator().manual_seed(100 + rank)          # its own shard
    return net, torch.randn(2, 6, 64, generator=g), torch.randint(13, (2, 64), generator=g)


rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
out = {"metric": "op-level"}
bench.attach_ddp_step(out, lambda: bench.ddp_step_measure(dist, rank, world, 3, 1, 2, 64, None, make=make))
if rank == 0:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)
dist.barrier()
dist.destroy_process_group()

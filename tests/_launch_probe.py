"""Rank body for tests/test_launch_cpu.py: what bench.py's ranks do around the timed region, on gloo."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from cloud_transformers_amd import parallel as P


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P.barrier(dist)
    dt = P.max_over_ranks(dist, 0.25 * (rank + 1))          # the slowest rank sets the job's time
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "world_size_seen": dist.get_world_size(), "dt": dt, "sum": float(t),
                          "local_rank": os.environ["LOCAL_RANK"], "addr": os.environ["MASTER_ADDR"]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

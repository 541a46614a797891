"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards over independent clouds (batch), so the data path needs no
collective; what the training step around it needs is the reference's small set
of helpers (utils/train_util_distributed.py:12-103) plus the benchmark's barrier
and max-over-ranks timing.  Everything here also runs on gloo (CPU) so that the
N>1 logic is covered by world_size-2 tests without a GPU.
"""
import torch


def _active(dist):
    return dist is not None and dist.is_available() and dist.is_initialized()


def world_size(dist):
    return dist.get_world_size() if _active(dist) else 1


def barrier(dist):
    if _active(dist):
        dist.barrier()


def _device_for(dist):
    if _active(dist) and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def max_over_ranks(dist, value):
    """Largest `value` (a python float) over all ranks; identity without a group."""
    if not _active(dist):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_range(total, rank, world):
    """Contiguous [begin, end) share of `total` independent units (clouds) for `rank`:
    sizes differ by at most one, every unit is owned exactly once."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_loss_dict(dist, loss_dict):
    """Sum-reduce a dict of scalar loss tensors to rank 0 and average them there
    (utils/train_util_distributed.py:12-34).  One stacked reduce instead of one per key."""
    if world_size(dist) < 2:
        return loss_dict
    with torch.no_grad():
        keys = sorted(loss_dict.keys())
        stacked = torch.stack([loss_dict[k].detach().reshape(()) for k in keys])
        dist.reduce(stacked, dst=0)
        if dist.get_rank() == 0:
            stacked = stacked / dist.get_world_size()
        return {k: v for k, v in zip(keys, stacked)}


def all_gather_tensor(dist, t):
    """Gather equally-shaped tensors from every rank (replaces the reference's
    pickle-through-byte-tensors all_gather, train_util_distributed.py:37-77, for the
    tensor payloads the training loops actually send)."""
    if world_size(dist) < 2:
        return [t]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.contiguous())
    return out

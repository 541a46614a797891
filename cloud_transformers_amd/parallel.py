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


def quiesce(seconds=0.5):
    """Before a HIP-graph capture that contains collectives: the device is idle (the caller synchronised), but the process
    group's watchdog thread still holds the finished work items of the eager warm-up until its next poll (every 100 ms) and
    queries their events then — which HIP refuses while the streams those events live on are capturing ("operation not
    permitted on an event last recorded in a capturing stream": the watchdog then takes the process down).  Waiting out a
    few polls lets it drop them first."""
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    time.sleep(seconds)


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


def all_gather(dist, data):
    """Gather one picklable object per rank -> list ordered by rank
    (utils/train_util_distributed.py:37-77; the reference pads pickled byte tensors by hand,
    torch.distributed.all_gather_object does the same exchange)."""
    if world_size(dist) < 2:
        return [data]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, data)
    return out


def data_parallel(model, device_index, sync_bn=True, **ddp_kwargs):
    """The reference's wrap (train_segmentation.py:128-130: SyncBatchNorm.convert_sync_batchnorm, then
    DistributedDataParallel on the rank's device) with the settings that matter on RCCL over xGMI:
    * gradients are views into the all-reduce buckets (no copy into and out of them);
    * under SyncBatchNorm the module buffers are not re-broadcast from rank 0 before every forward — every rank computes
      the same running statistics from the exchanged batch statistics, and the ~180 per-step broadcasts of a 12-block
      network are pure latency (one MI355X, world size 1: 35-40 ms per step with them, 30-34 ms without).  With plain
      BatchNorm (`sync_bn=False`) each rank's running statistics follow its own shard, so DDP's default re-broadcast
      stays on — otherwise the ranks' buffers drift apart silently and a checkpoint holds rank 0's shard statistics;
    * the converted norms keep the fused kernels: the blocks' norm groups exchange their statistics in ONE all_gather
      (forward) / ONE all_reduce (backward) per group (ops._bn_group_fwd / _bn_group_bwd), 6 collectives per
      MultiHeadUnion block and step instead of 2 per norm.
    `device_index=None` wraps a CPU module (gloo)."""
    import torch
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    kw = dict(device_ids=None if device_index is None else [device_index], gradient_as_bucket_view=True,
              broadcast_buffers=not sync_bn)
    kw.update(ddp_kwargs)
    return torch.nn.parallel.DistributedDataParallel(model, **kw)


def _plain_module(obj):
    return obj.module if isinstance(obj, torch.nn.parallel.DistributedDataParallel) else obj


def save_exp_parallel(objects, names, exp_path, epoch, epoch_name="epoch"):
    """`<exp_path>/<name>_<epoch_name>_<epoch>.t7` per object; DDP wrappers are saved unwrapped so the
    files load into a bare module (utils/train_util_distributed.py:80-88)."""
    assert len(objects) == len(names)
    for obj, name in zip(objects, names):
        with open("{}/{}_{}_{}.t7".format(str(exp_path), name, epoch_name, epoch), "wb") as f:
            torch.save(_plain_module(obj).state_dict(), f)


def restore_exp(dist, objects, names, device=None, verbose=True):
    """Load state dicts from the given paths on every rank, then barrier
    (utils/train_util_distributed.py:91-103)."""
    from pathlib import Path
    assert len(objects) == len(names)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    for obj, name in zip(objects, names):
        assert Path(name).exists()
        if verbose:
            print("restoring form {}".format(name))
        with open(name, "rb") as f:
            obj.load_state_dict(torch.load(f, map_location=device))
    barrier(dist)

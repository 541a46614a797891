"""`utils.train_util_distributed` with the reference's signatures (utils/train_util_distributed.py:12-103),
bound to the default torch.distributed group (RCCL on ROCm)."""
import torch
import torch.distributed as dist

from cloud_transformers_amd import parallel as _p


def reduce_loss_dict(loss_dict):
    return _p.reduce_loss_dict(dist, loss_dict)


def all_gather(data):
    return _p.all_gather(dist, data)


save_exp_parallel = _p.save_exp_parallel


def restore_exp(objects, names, device=torch.device('cuda:0'), verbose=True):
    return _p.restore_exp(dist, objects, names, device=device, verbose=verbose)

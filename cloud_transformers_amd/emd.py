"""Approximate Earth Mover's Distance (auction algorithm) on MI355X — counterpart of
the reference's `emd_linear/emd_module.py` (`emdFunction` :29-70, `emdModule` :72-77)
with the pybind11 `emd.forward/backward` calls (emd.cpp:28-31) replaced by the C ABI
entry points ct_emd_fwd / ct_emd_bwd.

Inputs: xyz1 (prediction), xyz2 (ground truth), both [B, n, 3] normalised to [0, 1];
n a multiple of 1024, B <= 512.  Returns (dist [B,n] squared distance to the assigned
target, assignment [B,n] int32).  Only xyz1 receives a gradient; the gradient wrt xyz2
is zeros, as in the reference (emd_module.py:66-70).  The twelve scratch tensors the
reference allocates per call (emd_module.py:41-54) are one opaque workspace here, and
tensors stay on the device of the inputs (the reference hard-codes 'cuda').
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib
from .ops import _dev, _ptr, _stream, _on


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert n == m
        assert xyz1.size()[0] == xyz2.size()[0]
        assert n % 1024 == 0
        assert batchsize <= 512
        _dev(xyz1, xyz2)
        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        dev = xyz1.device
        dist = torch.empty(batchsize, n, device=dev, dtype=torch.float32)
        assignment = torch.empty(batchsize, n, device=dev, dtype=torch.int32)
        lib = _lib.load()
        nws = lib.ct_emd_workspace_bytes(batchsize, n)
        ws = torch.empty(nws, device=dev, dtype=torch.uint8)
        with _on(dev):
            _lib.check(lib.ct_emd_fwd(_ptr(xyz1), _ptr(xyz2), _ptr(dist), _ptr(assignment), _ptr(ws), nws,
                                      batchsize, n, float(eps), int(iters), _stream()), "ct_emd_fwd")
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        graddist = graddist.contiguous()
        gradxyz1 = torch.empty_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        lib = _lib.load()
        B, n, _ = xyz1.shape
        with _on(xyz1.device):
            _lib.check(lib.ct_emd_bwd(_ptr(xyz1), _ptr(xyz2), _ptr(graddist), _ptr(assignment), _ptr(gradxyz1),
                                      B, n, _stream()), "ct_emd_bwd")
        return gradxyz1, gradxyz2, None, None


class emdModule(nn.Module):
    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)

"""Point-wise (kernel_size 1) Conv1d of the MHCT blocks — `keys_values_pred`, the union's `after` and
`shortcut` projections (reference layers/multihead_ct.py:31-33,149-160).

A plain library GEMM, and it stays one: forward and the data gradient go through torch's conv1d (MIOpen
picks rocBLAS/Tensile GEMMs for them).  For the WEIGHT gradient MIOpen picks an NHWC implicit-GEMM
kernel bracketed by layout transposes of x and g_y; `g_w = sum_b g_y[b] @ x[b]^T` as one rocBLAS batched
GEMM on the tensors as they lie needs no transposes and is 18-24 % faster on the whole fwd+bwd of these
layers at B8 N4096 (tools/conv1d_bench.py).  `PointwiseConv1d` subclasses nn.Conv1d: same parameters,
same state-dict keys, same results.
"""
import torch
import torch.nn.functional as F
from torch import nn


class _PointwiseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.conv1d(x, weight, bias)

    @staticmethod
    def backward(ctx, g_y):
        x, weight = ctx.saved_tensors
        g_y = g_y.contiguous()
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = F.conv1d(g_y, weight.transpose(0, 1).contiguous())
        if ctx.needs_input_grad[1]:
            g_w = torch.bmm(g_y, x.transpose(1, 2)).sum(0).unsqueeze(-1)      # [O, I, 1]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g_b = g_y.sum(dim=(0, 2))
        return g_x, g_w, g_b


class PointwiseConv1d(nn.Conv1d):
    def _eligible(self, x):
        return (x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and self.kernel_size == (1,)
                and self.stride == (1,) and self.padding == (0,) and self.dilation == (1,) and self.groups == 1
                and self.padding_mode == "zeros")

    def forward(self, x):
        if self._eligible(x):
            return _PointwiseConvFn.apply(x, self.weight, self.bias)
        return super().forward(x)

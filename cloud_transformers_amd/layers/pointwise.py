"""Point-wise (kernel_size 1) Conv1d of the MHCT blocks — `keys_values_pred`, the union's `after` and
`shortcut` projections (reference layers/multihead_ct.py:31-33,149-160).

A plain library GEMM, and it stays one — three rocBLAS batched GEMMs on the tensors as they lie:
`y[b] = W @ x[b]`, `g_x[b] = W^T @ g_y[b]` (W and W^T as broadcast views), `g_w = sum_b g_y[b] @ x[b]^T`.
torch's conv1d reaches the same GEMMs through MIOpen for forward / data gradient (5-10 % slower at these
shapes), but for the WEIGHT gradient MIOpen picks an NHWC implicit-GEMM kernel bracketed by layout
transposes of x and g_y: 18-24 % slower on the whole fwd+bwd of these layers at B8 N4096
(tools/conv1d_bench.py).
`PointwiseConv1d` subclasses nn.Conv1d: same parameters,
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
        w2 = weight[:, :, 0]
        # bmm with W broadcast over the batch keeps the [B, O, N] layout (torch.matmul(W, x) folds the batch into
        # the GEMM's M and returns a transposed view, which costs transposing copies downstream)
        y = torch.bmm(w2.unsqueeze(0).expand(x.size(0), -1, -1), x)
        return y if bias is None else y + bias[None, :, None]

    @staticmethod
    def backward(ctx, g_y):
        x, weight = ctx.saved_tensors
        g_y = g_y.contiguous()
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = torch.bmm(weight[:, :, 0].t().unsqueeze(0).expand(g_y.size(0), -1, -1), g_y)    # W^T as a view
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

"""Point-wise (kernel_size 1) Conv1d of the MHCT blocks — `keys_values_pred`, the union's `after` and
`shortcut` projections (reference layers/multihead_ct.py:31-33,149-160).

Three products on the tensors as they lie — `y[b] = W @ x[b]`, `g_x[b] = W^T @ g_y[b]`, `g_w = sum_b g_y[b] @ x[b]^T` —
through `ops.pw_forward` / `ops.pw_backward`: this library's split-f16 MFMA kernels (csrc/ct_pwgemm.hip: fp32 in and out,
22-bit operands, fp32 accumulation) where the sizes are multiples of 4, rocBLAS fp32 batched GEMMs (torch.bmm, W and W^T
as broadcast views) otherwise or under CLOUDCT_PW_GEMM=lib.  torch's own conv1d reaches the library GEMMs through MIOpen for
forward / data gradient and an NHWC implicit-GEMM kernel bracketed by layout transposes for the weight gradient
(tools/conv1d_bench.py, tools/pw_gemm_bench.py).
`PointwiseConv1d` subclasses nn.Conv1d: same parameters,
same state-dict keys, same results.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops


class _PointwiseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        w2 = weight[:, :, 0].contiguous()
        y, am_w, am_x, wt = ops.pw_forward(w2, x, ctx.needs_input_grad[0])
        am_w = am_w if am_w is not None else (None, None)       # (row maxima, column maxima) of the weight
        ctx.save_for_backward(x, w2, am_w[0], am_w[1], am_x, wt)
        ctx.has_bias = bias is not None
        return y if bias is None else y + bias[None, :, None]

    @staticmethod
    def backward(ctx, g_y):
        x, w2, am_wr, am_wc, am_x, wt = ctx.saved_tensors
        g_y = g_y.contiguous()
        g_b = None
        g_x, g_w = ops.pw_backward(w2, x, g_y, None if am_wr is None else (am_wr, am_wc), am_x, ctx.needs_input_grad[0],
                                   ctx.needs_input_grad[1], Wt=wt)
        if g_w is not None:
            g_w = g_w.unsqueeze(-1)                                           # [O, I, 1]
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


def convert_pointwise(module):
    """Turn every plain `nn.Conv1d` of `module` that is a pointwise convolution (kernel size 1, stride 1, no padding /
    dilation / groups) into a `PointwiseConv1d`, in place — the stems and heads the reference's model files build with
    `nn.Conv1d(C, C', 1)` (model_zoo/s3dis/segmenter.py, model_zoo/completion/inpainter.py) then take the same kernels as the
    blocks' own projections.  The class has no state of its own: parameters, buffers, hooks and state-dict keys stay as they
    are (the counterpart of `SyncBatchNorm.convert_sync_batchnorm` for this layer).  Returns `module`."""
    for m in module.modules():
        if type(m) is nn.Conv1d and m.kernel_size == (1,) and m.stride == (1,) and m.padding == (0,) and m.dilation == (1,) \
                and m.groups == 1 and m.padding_mode == "zeros":
            m.__class__ = PointwiseConv1d
    return module

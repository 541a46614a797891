"""Grouped 3^d convolution modules backed by the MFMA kernels of csrc/ct_gconv.hip.

`GroupedConv2d` / `GroupedConv3d` subclass `nn.Conv2d` / `nn.Conv3d`, so parameter names,
shapes, initialisation and state dicts are exactly those of the `nn.Conv{2,3}d(...,
groups=heads)` layers the reference builds (layers/multihead_ct.py:50-65,
unet2d/unet_parts.py:13-16, layers/v2v_groups.py:26-29).  The forward/backward run on the
hand-written kernels whenever the layer is the shape the MHCT path uses — kernel 3, stride 1,
padding 1, dilation 1, zero padding, fp32 on a HIP device.  The grouped 1x1 skip projections of the Res stacks
(layers/v2v_groups.py:40-44) are one batched GEMM per (cloud, group) on the BLAS library — the library's grouped
convolution spends ~0.4 ms per weight gradient on these few-kFLOP layers; any other configuration goes to the
stock PyTorch/MIOpen implementation of the parent class.
"""
import torch
from torch import nn

from .. import _lib
from ..ops import _ptr, _stream, _on


class GroupedConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups):
        x = x.contiguous()
        weight = weight.contiguous()
        dim = x.dim() - 2
        B = x.shape[0]
        Cin = x.shape[1] // groups
        Cout = weight.shape[0] // groups
        W = list(x.shape[2:])
        y = torch.empty(B, groups * Cout, *W, device=x.device, dtype=torch.float32)
        lib = _lib.load()
        with _on(x.device):
            _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(weight), _ptr(bias.contiguous()) if bias is not None else None,
                                        _ptr(y), B, groups, Cin, Cout, dim, _lib.int_array(W), _stream()),
                       "ct_gconv_fwd")
        ctx.save_for_backward(x, weight)
        ctx.meta = (groups, Cin, Cout, dim, W, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, weight = ctx.saved_tensors
        groups, Cin, Cout, dim, W, has_bias = ctx.meta
        g_y = g_y.contiguous()
        B = x.shape[0]
        lib = _lib.load()
        Wa = _lib.int_array(W)
        g_x = g_w = g_b = None
        with _on(x.device):
            if ctx.needs_input_grad[0]:
                g_x = torch.empty_like(x)
                _lib.check(lib.ct_gconv_bwd_data(_ptr(g_y), _ptr(weight), _ptr(g_x), B, groups, Cin, Cout, dim, Wa,
                                                 _stream()), "ct_gconv_bwd_data")
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                g_w = torch.empty_like(weight)
                g_b = torch.empty(groups * Cout, device=x.device, dtype=torch.float32) if has_bias else None
                ws_bytes = lib.ct_gconv_bwd_weight_workspace_bytes(B, groups, Cin, Cout, dim, Wa)
                ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8) if ws_bytes else None
                _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(g_y), _ptr(g_w), _ptr(g_b), _ptr(ws), ws_bytes,
                                                   B, groups, Cin, Cout, dim, Wa, _stream()), "ct_gconv_bwd_weight")
        return g_x, g_w, g_b, None


def _eligible(mod, x):
    nd = x.dim() - 2
    if not (x.is_cuda and x.dtype == torch.float32 and mod.weight.dtype == torch.float32
            and tuple(mod.kernel_size) == (3,) * nd and tuple(mod.stride) == (1,) * nd
            and tuple(mod.padding) == (1,) * nd and tuple(mod.dilation) == (1,) * nd
            and mod.padding_mode == "zeros"):
        return False
    # More than 32 channels per group (the grouped Res2D / Res3D stacks on the pooled 8^3 .. 2^3 volumes and 16^2 .. 4^2
    # planes of the classifier / inpainter encoders) run on the K-split MFMA kernel (gconv_fwd4k_kernel: contraction in
    # blocks of 16 input channels) and the small-volume weight-gradient kernel; measured fwd+bwd at B8, groups 16, vs the
    # library: 3D 1024->1024 8^3 0.71 vs 1.46 ms, 512->1024 8^3 0.42 vs 0.90, 4^3 0.22 vs 0.46, 2D 8^2 0.10 vs 0.13,
    # 4^2 0.08 vs 0.13; 2^d volumes have their own dense kernel (gconv_tiny_kernel).  One exception: other rows that are
    # not 16-byte multiples with wide groups — the quad MFMA kernels need float4 rows and the one-position form loses to
    # the library there.
    if (max(mod.in_channels, mod.out_channels) // mod.groups > 32 and x.shape[-1] % 4 != 0
            and tuple(x.shape[2:]) != (2,) * nd):
        return False
    # shapes whose tiles do not fit LDS (very wide rows with many channels per group) take the library convolution
    W = tuple(x.shape[2:])
    key = (x.shape[0], mod.groups, mod.in_channels // mod.groups, mod.out_channels // mod.groups, W)
    ok = _SUPPORTED.get(key)
    if ok is None:
        ok = bool(_lib.load().ct_gconv_supported(key[0], key[1], key[2], key[3], nd, _lib.int_array(W)))
        _SUPPORTED[key] = ok
    return ok


_SUPPORTED = {}


def _pointwise(mod, x):
    nd = x.dim() - 2
    return (x.is_cuda and x.dtype == torch.float32 and mod.weight.dtype == torch.float32
            and tuple(mod.kernel_size) == (1,) * nd and tuple(mod.stride) == (1,) * nd
            and tuple(mod.padding) == (0,) * nd and tuple(mod.dilation) == (1,) * nd)


def _pointwise_grouped(mod, x):
    """y[b, g, :, p] = W[g] @ x[b, g, :, p] (+ bias): batch of groups x clouds plain GEMMs; autograd supplies the two
    transposed products of the backward."""
    B, G = x.shape[0], mod.groups
    sp = x.shape[2:]
    co, ci = mod.out_channels // G, mod.in_channels // G
    y = torch.matmul(mod.weight.reshape(1, G, co, ci), x.reshape(B, G, ci, -1)).reshape(B, G * co, *sp)
    if mod.bias is not None:
        y = y + mod.bias.reshape(1, -1, *([1] * len(sp)))
    return y


class GroupedConv2d(nn.Conv2d):
    def forward(self, x):
        if _eligible(self, x):
            return GroupedConvFn.apply(x, self.weight, self.bias, self.groups)
        if _pointwise(self, x):
            return _pointwise_grouped(self, x)
        return super().forward(x)


class GroupedConv3d(nn.Conv3d):
    def forward(self, x):
        if _eligible(self, x):
            return GroupedConvFn.apply(x, self.weight, self.bias, self.groups)
        if _pointwise(self, x):
            return _pointwise_grouped(self, x)
        return super().forward(x)

"""Per-head rigid transforms and AdaIN for the MHCT blocks.

Counterparts of the reference's `layers/utils.py`: `VolTransformer` (:9-34),
`PlaneTransformer` (:37-61), `AdaIn1dUpd` (:82-97), `forward_stats` (:64-79).
Parameter names (`log_R`, `shift`, `scales`, `linear.*`) and shapes are the
reference's, so released checkpoints load with strict=True.  The corner-weight
helpers `bilinear_coords` / `trilinear_coords` (:100-186) live inside the HIP
kernels (csrc/ct_common.h) and are exposed here only as thin wrappers over
`DifferentiablePositions`' kernel for API completeness.
"""
import torch
from torch import nn

from .. import ops


def so3_exponential_map(log_rot, eps=1e-4):
    """Rodrigues' formula, the map pytorch3d.transforms.so3.so3_exponential_map
    publishes (the reference imports it, layers/utils.py:6): for v in R^3 with
    theta = sqrt(clamp(|v|^2, eps)),  R = I + (sin theta / theta) K + ((1 - cos theta) / theta^2) K^2,
    K = hat(v).  log_rot [H,3] -> [H,3,3]."""
    assert log_rot.dim() == 2 and log_rot.size(1) == 3
    if log_rot.dtype == torch.float32:
        return ops.so3_exp(log_rot, eps)                  # one HIP launch each way instead of ~40 tiny ones; raises off the GPU
    if not log_rot.is_cuda:
        raise RuntimeError("cloud_transformers_amd has no CPU fallback: so3_exponential_map needs a HIP tensor")
    # other floating types on the GPU (float64 checks): the same formula in torch ops
    sq = (log_rot * log_rot).sum(dim=1)
    theta = sq.clamp(min=eps).sqrt()
    a = theta.sin() / theta
    b = (1.0 - theta.cos()) / (theta * theta)
    x, y, z = log_rot.unbind(dim=1)
    o = torch.zeros_like(x)
    K = torch.stack((o, -z, y,
                     z, o, -x,
                     -y, x, o), dim=1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device).expand_as(K)
    return eye + a.view(-1, 1, 1) * K + b.view(-1, 1, 1) * torch.bmm(K, K)


class _RigidTransformer(nn.Module):
    """pcd [B,H,3,N] -> (pcd + shift) as ROW vectors times R(log_R) per head,
    optionally scaled; subclasses choose how many rotated coordinates are kept."""

    out_dims = 3

    def __init__(self, heads, scales=False):
        super().__init__()
        self.heads = heads
        self.log_R = nn.Parameter(torch.randn(heads, 3, dtype=torch.float32))
        self.shift = nn.Parameter(torch.zeros(heads, 3, dtype=torch.float32))
        self.do_scales = scales
        if scales:
            self.scales = nn.Parameter(torch.ones(heads, self.out_dims, dtype=torch.float32))

    def forward(self, pcd):
        R = so3_exponential_map(self.log_R)                       # [H,3(c),3(n)]
        moved = pcd + self.shift[None, :, :, None]
        # out[b,h,n,p] = sum_c moved[b,h,c,p] * R[h,c,n]
        out = torch.matmul(R.transpose(1, 2)[None], moved)
        out = out[:, :, :self.out_dims]
        if self.do_scales:
            out = out * self.scales[None, :, :, None]
        return out


class VolTransformer(_RigidTransformer):
    out_dims = 3


class PlaneTransformer(_RigidTransformer):
    out_dims = 2          # planes keep the first two rotated coordinates (utils.py:58-61)


class AdaIn1dUpd(nn.Module):
    """InstanceNorm1d (no affine) followed by a style-predicted scale (+1) and bias.
    The class NAME is part of the interface: `forward_style` dispatches on it
    (layers/multihead_ct_adain.py:11)."""

    def __init__(self, num_features, num_latent):
        super().__init__()
        self.num_features = num_features
        self.num_latent = num_latent
        self.instance_norm = nn.InstanceNorm1d(num_features, eps=1e-5, affine=False)
        self.linear = nn.Linear(num_latent, num_features * 2)

    _gb = None     # set by a union block for the span of its forward: this norm's slice of the block's stacked style projection

    def gamma_beta(self, z):
        """[B,2,C] scale / bias predicted from the style vector: the block's stacked projection when one is in flight
        (MultiHeadUnionAdaIn.forward, ops.StyleProjFn), else this layer's own Linear."""
        if self._gb is not None:
            return self._gb
        return self.linear(z).reshape(-1, 2, self.num_features)

    def forward(self, x, z, relu=False, residual=None):
        """`relu=True` folds the ReLU that follows this layer in the blocks' `after` stacks into the same
        kernel (forward_style passes it and skips the nn.ReLU); `residual` is added to the result in the same pass."""
        gamma_beta = self.gamma_beta(z)
        if x.dtype == torch.float32 and x.dim() == 3:
            return ops.adain(x, gamma_beta, self.instance_norm.eps, relu, residual)    # one HIP launch (ct_adain_fwd); raises off the GPU
        if not x.is_cuda:
            raise RuntimeError("cloud_transformers_amd has no CPU fallback: AdaIn1dUpd needs a HIP tensor")
        # other layouts / floating types on the GPU: torch's own composition
        y = self.instance_norm(x) * (gamma_beta[:, 0, :, None] + 1) + gamma_beta[:, 1, :, None]
        y = torch.relu(y) if relu else y
        return y if residual is None else y + residual


def forward_stats(input, module, type):
    """Run a Sequential whose layers of class `type` return (output, lattice stats);
    collect the stats (reference layers/utils.py:64-79)."""
    collected = []
    x = input
    for layer in module:
        if isinstance(layer, type):
            x, st = layer(x)
            collected.extend(st if isinstance(st, list) else [st])
        else:
            x = layer(x)
    return x, collected


def _coords(keys_scaled, dim):
    # keys_scaled: [B', dim, N] already in grid units [0, W-1); weights/cells via the same
    # device math as the kernels: floor, low/high weights, corner order v = dx + 2dy (+4dz)
    f = keys_scaled.floor()
    lo, hi = (f + 1) - keys_scaled, keys_scaled - f
    ws, cells = [], []
    for v in range(1 << dim):
        w = None
        off = []
        for j in range(dim):
            bit = (v >> j) & 1
            wj = hi[:, j] if bit else lo[:, j]
            w = wj if w is None else w * wj
            off.append(f[:, j].long() + bit)
        ws.append(w)
        cells.append(torch.stack(off, dim=1))
    return torch.stack(ws, dim=1), torch.stack(cells, dim=1)


def bilinear_coords(keys):
    """[B',2,N] grid-unit keys -> (weights [B',4,N], integer corner coords [B',4,2,N])."""
    assert keys.shape[1] == 2
    return _coords(keys, 2)


def trilinear_coords(keys):
    """[B',3,N] grid-unit keys -> (weights [B',8,N], integer corner coords [B',8,3,N])."""
    assert keys.shape[1] == 3
    return _coords(keys, 3)

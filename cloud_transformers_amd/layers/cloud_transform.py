"""Core differentiable rasterize / de-rasterize modules on MI355X.

Drop-in counterparts of the reference's `layers/cloud_transform.py`
(`DifferentiablePositions` :62-121, `Splat` :124-180, `Slice` :183-227): same
constructor arguments, same forward signatures, same `tensor_mod` buffer in the
state dict (:48-51) — but every forward/backward runs in hand-written HIP
kernels behind the C ABI of include/cloudct.h.

Besides the reference-compatible `forward(local_coordinate, flattened_index, …)`
each of `Splat` / `Slice` offers `forward_keys(keys, …)`: the fused hot path
used by the MultiHead* blocks, which recomputes the corner weights/indices
inside the kernels and never materialises `local_coordinate` /
`flattened_index` (or the reference's (B,H,C,V,N) intermediates) in HBM.

Code written the reference's way — `lc, idx = positions(keys); z = splat(lc, idx, f);
o = slice(lc, idx, conv(z))` (layers/multihead_ct.py:99-107) — takes the same fused
path: `DifferentiablePositions.forward` tags the pair it returns with the keys it
came from, and `Splat` / `Slice` given exactly that untouched pair (same tensor
objects, no in-place edit since, same grid) run `forward_keys` on those keys.  The
value and the gradient wrt the keys are those of the explicit path (the kernels
recompute the very same weights and indices); a pair that was detached, cloned,
sliced or edited is a different tensor and takes the explicit `(lc, idx)` kernels.
"""
import torch
from torch import nn

from .. import ops


class DifferentiableGridModule(nn.Module):
    def __init__(self, tensor_size=20, heads=4, dim=3):
        """
        :param tensor_size: spatial resolution of the feature map; int, or tuple with len() == dim
        :param heads: number of parallel de/rasterizations, int > 0
        :param dim: 2 or 3
        """
        super().__init__()
        assert dim in (2, 3)
        self.dim = dim
        self.heads = heads
        if isinstance(tensor_size, int):
            self.tensor_size = dim * [tensor_size]
        else:
            assert isinstance(tensor_size, (tuple, list))
            assert len(tensor_size) == dim
            self.tensor_size = [int(w) for w in tensor_size]
        # kept for state-dict compatibility with released checkpoints
        self.register_buffer("tensor_mod",
                             torch.tensor(self.tensor_size, dtype=torch.float32)[None, :, None])
        self.spread_size = 1 << dim
        self.eps = 1e-7


class DifferentiablePositions(DifferentiableGridModule):
    """keys f32[B, heads*dim, N] -> (local_coordinate f32[B,heads,V,N], flattened_index i64[B,heads,V,N])."""

    def forward(self, keys):
        assert keys.size(1) == self.heads * self.dim
        lc, idx = ops.positions(keys, self.tensor_size, self.heads, self.dim)
        if keys.is_cuda and keys.dtype == torch.float32:
            tag = _Provenance(keys, self.tensor_size, self.heads, self.dim, lc, idx)
            lc._ct_src = tag
            idx._ct_src = tag
        return lc, idx


class _Provenance:
    """Where a (local_coordinate, flattened_index) pair came from: lets Splat / Slice run the fused keys path."""
    __slots__ = ("keys", "config", "versions")

    def __init__(self, keys, tensor_size, heads, dim, lc, idx):
        self.keys = keys
        self.config = (tuple(tensor_size), heads, dim)
        self.versions = (keys._version, lc._version, idx._version)

    def keys_for(self, module, lc, idx):
        """The keys if (lc, idx) is the untouched pair this tag was made for and `module` has the same grid, else None."""
        if getattr(idx, "_ct_src", None) is not self:
            return None
        if (tuple(module.tensor_size), module.heads, module.dim) != self.config:
            return None
        if (self.keys._version, lc._version, idx._version) != self.versions:
            return None
        return self.keys


class Splat(DifferentiableGridModule):
    """Differentiable rasterization into a 2D/3D feature grid.

    reduce="max" is what the reference executes (scatter_max into zeros,
    layers/cloud_transform.py:164-173); reduce="sum" is the scatter-add variant.
    """

    def __init__(self, tensor_size=20, heads=4, dim=3, reduce="max"):
        super().__init__(tensor_size, heads, dim)
        assert reduce in ("max", "sum")
        self.reduce = reduce

    def forward(self, local_coordinate, flattened_index, features, pts_padding=None):
        assert features.dtype == torch.float32
        assert features.size(1) % self.heads == 0
        tag = getattr(local_coordinate, "_ct_src", None)
        keys = tag.keys_for(self, local_coordinate, flattened_index) if tag is not None else None
        if keys is not None:
            return self.forward_keys(keys, features, pts_padding)
        return ops.splat_lc(local_coordinate, flattened_index, features, pts_padding,
                            self.tensor_size, self.heads, self.dim, self.reduce)

    def forward_keys(self, keys, features, pts_padding=None):
        assert features.dtype == torch.float32
        assert features.size(1) % self.heads == 0
        return ops.splat_keys(keys, features, pts_padding, self.tensor_size, self.heads, self.dim, self.reduce)


class Slice(DifferentiableGridModule):
    """Differentiable sampling of a 2D/3D feature grid back to the points."""

    def forward(self, local_coordinate, flattened_index, convolved, pts_padding=None):
        assert convolved.size(1) % self.heads == 0
        tag = getattr(local_coordinate, "_ct_src", None)
        keys = tag.keys_for(self, local_coordinate, flattened_index) if tag is not None else None
        if keys is not None:
            return self.forward_keys(keys, convolved, pts_padding)
        return ops.slice_lc(local_coordinate, flattened_index, convolved, pts_padding,
                            self.tensor_size, self.heads, self.dim)

    def forward_keys(self, keys, convolved, pts_padding=None):
        assert convolved.size(1) % self.heads == 0
        return ops.slice_keys(keys, convolved, pts_padding, self.tensor_size, self.heads, self.dim)

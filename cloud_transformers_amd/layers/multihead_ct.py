"""Multi-Headed Cloud Transform blocks on MI355X.

Counterparts of the reference's `layers/multihead_ct.py` (`MultiHead` :9-118,
`MultiHeadUnion` :121-198), `layers/multihead_ct_adain.py` (`forward_style`
:8-16, `MultiHeadAdaIn` :19-136, `MultiHeadUnionAdaIn` :139-218) and
`layers/multihead_ct_pool.py` (`MultiHeadPool` :9-86): same constructor
arguments, forward signatures, return contract `(result, stats)` and the same
sub-module / parameter / buffer NAMES, so reference checkpoints load with
strict=True and reference-style model_zoo files can use these classes unchanged.

What differs is what happens between the pointwise convolutions: the
rasterize -> grouped conv -> de-rasterize core runs on the fused HIP path
(`Splat.forward_keys` / `Slice.forward_keys`): corner weights and cell indices
are recomputed in-kernel from the lattice, so `local_coordinate`,
`flattened_index` and the reference's (B,H,C,V,N) intermediates never exist in
HBM, there are no host synchronisations (the reference's four `.all()` asserts
per block, cloud_transform.py:101-111), and the occupancy statistic is one small
device reduction.
"""
import os

import torch
from torch import nn

from .. import ops
from .cloud_transform import DifferentiablePositions, Slice, Splat
from .gconv import GroupedConv2d, GroupedConv3d
from .pointwise import PointwiseConv1d
from .utils import AdaIn1dUpd, PlaneTransformer, VolTransformer, so3_exponential_map


LATTICE_SO3 = os.environ.get("CLOUDCT_LATTICE_SO3", "1") != "0"      # "0": so3 map and lattice as separate ops (A/B)


def forward_style(module_list, input, z, residual=None):
    """Apply a Sequential in which AdaIN layers also take the style vector `z` (and add `residual` to the result).
    Dispatch is by class name, like the reference (multihead_ct_adain.py:11)."""
    layers = list(module_list)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if "AdaIn1dUpd" in str(type(layer)):
            fuse = i + 1 < len(layers) and type(layers[i + 1]) is nn.ReLU and isinstance(layer, AdaIn1dUpd)
            step = 2 if fuse else 1
            last = i + step == len(layers)
            fuse_res = (last and residual is not None and isinstance(layer, AdaIn1dUpd) and residual.shape == input.shape
                        and residual.dtype == input.dtype)
            if fuse_res:
                input = layer(input, z, relu=fuse, residual=residual)
                residual = None
            else:
                input = layer(input, z, relu=True) if fuse else layer(input, z)
            i += step
        else:
            input = layer(input)
            i += 1
    return input if residual is None else residual + input


def run_after(module_list, input, residual=None):
    """Apply an `after` Sequential (and add `residual` to its result); a training-mode nn.BatchNorm1d directly followed by
    nn.ReLU runs as the fused ct_bn_relu kernels when the shape qualifies (ops.bn_relu_eligible), the skip connection
    added in the same pass when that pair ends the stack (an nn.SyncBatchNorm too: ops exchanges the group's statistics
    over its process group) — eval mode and every other layer go through their own forward."""
    layers = list(module_list)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if i + 1 < len(layers) and type(layers[i + 1]) is nn.ReLU and ops.bn_relu_eligible(layer, input):
            last = i + 2 == len(layers)
            fuse_res = last and residual is not None and residual.shape == input.shape and residual.dtype == input.dtype
            input = ops.bn_relu(input, layer, relu=True, residual=residual if fuse_res else None)
            if fuse_res:
                residual = None
            i += 2
        else:
            input = layer(input)
            i += 1
    return input if residual is None else residual + input


def _grouped_conv(tensor_dim, channels, heads):
    conv = GroupedConv3d if tensor_dim == 3 else GroupedConv2d      # nn.Conv{2,3}d subclasses on the MFMA kernels
    return nn.Sequential(conv(channels, channels, kernel_size=3, stride=1, padding=1, groups=heads, bias=True))


class _MHCTCore(nn.Module):
    """Shared plumbing of the three MHCT flavours: key/value prediction, per-head
    rigid transform, lattice, fused Splat / Slice and the lattice statistics."""

    def _build_core(self, model_dim, in_feature_dim, tensor_size, tensor_dim, heads, with_slice):
        assert tensor_dim == 3 or tensor_dim == 2
        self.in_feature_dim = in_feature_dim
        self.model_dim = model_dim
        self.tensor_size = tensor_size
        self.tensor_dim = tensor_dim
        self.heads = heads
        self.keys_values_pred = nn.Sequential(
            PointwiseConv1d(model_dim, heads * (in_feature_dim + 3), kernel_size=1, bias=False))

    def _build_grid_modules(self, with_slice):
        kw = dict(tensor_size=self.tensor_size, dim=self.tensor_dim, heads=self.heads)
        self.diff_poss = DifferentiablePositions(**kw)
        self.splat = Splat(**kw)
        if with_slice:
            self.slice = Slice(**kw)

    def _build_transform(self, scales):
        cls = VolTransformer if self.tensor_dim == 3 else PlaneTransformer
        self.transform = cls(self.heads, scales=scales)

    def _lattice(self, orig_pcd, keys_res, kscale=None):
        """keys = transform(xyz + kscale * residual) per head; lattice = tanh(keys) — one fused HIP
        kernel each way (ct_lattice_fwd / _bwd); only the H 3x3 rotations are built by torch."""
        t = self.transform
        if LATTICE_SO3 and t.log_R.dtype == torch.float32:
            # the so3 map inside the lattice launches: 1 launch forward, 2 backward (ops.LatticeSo3Fn)
            return ops.lattice_so3(orig_pcd, keys_res, t.log_R, t.shift, t.scales if t.do_scales else None, kscale, self.tensor_dim,
                                   with_stats=True)
        R = so3_exponential_map(t.log_R)
        return ops.lattice(orig_pcd, keys_res, R, t.shift, t.scales if t.do_scales else None, kscale, self.tensor_dim,
                           with_stats=True)

    def _norm_keys_values(self, key_values):
        """key_bn on the first 3H channels, values_bn on the rest (multihead_ct.py:89-91): fused kernels on the slices
        where they lie when both norms qualify, the modules on split views otherwise (eval mode, ...)."""
        Ck = self.heads * 3
        Cv = key_values.size(1) - Ck
        if (ops.bn_relu_eligible(self.key_bn, key_values, Ck) and ops.bn_relu_eligible(self.values_bn, key_values, Cv)
                and ops.norms_share_group([self.key_bn, self.values_bn])):
            return ops.split_bn(key_values, self.key_bn, self.values_bn)
        k_part, v_part = torch.split(key_values, [Ck, Cv], dim=1)   # backward: one cat
        return self.key_bn(k_part), self.values_bn(v_part)

    def _core(self, lattice, values, pts_padd=None):
        """(sliced features, occupancy) = Slice(conv(Splat(values))) at the lattice (multihead_ct.py:99-107).  The grids whose
        two tiles fit a CU run as ONE kernel that keeps them in LDS (ops.mhct_core: 32^2 / 16^2 with 16 features per head,
        8^3 with 32 — measured 1.15-1.7x the three-kernel chain on the forward, tools/core_bench.py); every other grid goes
        through Splat -> grouped conv -> Slice."""
        B = lattice.size(0)
        conv = self.conv[0] if len(self.conv) == 1 else None
        if (ops.FUSED_CORE and conv is not None and isinstance(conv, (GroupedConv2d, GroupedConv3d)) and lattice.is_cuda
                and values.dtype == torch.float32 and lattice.dtype == torch.float32 and lattice.size(2) % 4 == 0
                and tuple(conv.kernel_size) == (3,) * self.tensor_dim and tuple(conv.stride) == (1,) * self.tensor_dim
                and tuple(conv.padding) == (1,) * self.tensor_dim and tuple(conv.dilation) == (1,) * self.tensor_dim
                and conv.padding_mode == "zeros" and conv.groups == self.heads and conv.in_channels == conv.out_channels
                and conv.weight.dtype == torch.float32
                and ops.mhct_core_supported(B, self.heads, self.in_feature_dim, lattice.size(2), ops.sizes_of(self.tensor_size, self.tensor_dim))):
            pre, count = ops.mhct_core(lattice, values, pts_padd, conv.weight, conv.bias, self.tensor_size, self.heads, self.tensor_dim)
            with torch.no_grad():
                occ = count * ops.occupancy_scale(B * self.in_feature_dim * self.heads)      # (one launch: int64 x float scalar -> float32)
            return pre, occ
        z = self.splat.forward_keys(lattice, values, pts_padd)
        occ = self._occupancy(z, B)
        return self.slice.forward_keys(lattice, self.conv(z), pts_padd), occ

    def _occupancy(self, z, batch):
        with torch.no_grad():
            if z.numel() < 2 ** 31:
                return ops.grid_occupancy_ratio(z, batch * self.in_feature_dim * self.heads)
            return ops.grid_occupancy_count(z).float() / (batch * self.in_feature_dim * self.heads)


class MultiHead(_MHCTCore):
    def __init__(self, model_dim, in_feature_dim, out_model_dim, tensor_size, tensor_dim, heads, scales=False):
        super().__init__()
        self.out_model_dim = out_model_dim       # accepted, unused — as in the reference
        self._build_core(model_dim, in_feature_dim, tensor_size, tensor_dim, heads, True)
        self.values_bn = nn.BatchNorm1d(heads * in_feature_dim)
        self.key_bn = nn.BatchNorm1d(heads * 3)
        self._build_grid_modules(True)
        self.conv = _grouped_conv(tensor_dim, heads * in_feature_dim, heads)
        self.after = nn.Sequential(nn.BatchNorm1d(heads * in_feature_dim), nn.ReLU(inplace=True))
        self._build_transform(scales)
        self._reset_parameters()

    def _reset_parameters(self):
        # keys start as the pure rigid transform of xyz (multihead_ct.py:79-80)
        nn.init.zeros_(self.key_bn.weight)

    def _forward_pre(self, input, orig_pcd, kv=None):
        """Everything up to (not including) `after`: (sliced features, stats, lattice).  `kv` = (keys_res, values) when the
        union block computed the projection and its norms for all heads at once."""
        pts_padd = None
        if isinstance(orig_pcd, tuple):
            orig_pcd, pts_padd = orig_pcd
        if kv is None:
            keys_res, values = self._norm_keys_values(self.keys_values_pred(input))
        else:
            keys_res, values = kv
        keys, lattice, kstats = self._lattice(orig_pcd, keys_res)
        pre, occ = self._core(lattice, values, pts_padd)
        with torch.no_grad():
            stats = (occ, kstats[0], kstats[1], None)       # mean / variance of the keys, reduced by the lattice kernel
        return pre, stats, lattice

    def forward(self, input, orig_pcd, return_lattice=False):
        pre, stats, lattice = self._forward_pre(input, orig_pcd)
        result = run_after(self.after, pre)
        if return_lattice:
            result = result, lattice
        return result, stats


class MultiHeadPool(_MHCTCore):
    """Splat-only MHCT: a learned pooling of the cloud into 2D/3D grids."""

    def __init__(self, model_dim, in_feature_dim, tensor_size, tensor_dim, heads, scales=False):
        super().__init__()
        self._build_core(model_dim, in_feature_dim, tensor_size, tensor_dim, heads, False)
        self.values_bn = nn.BatchNorm1d(heads * in_feature_dim)
        self.key_bn = nn.BatchNorm1d(heads * 3)
        self._build_grid_modules(False)
        self._build_transform(scales)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.zeros_(self.key_bn.weight)

    def forward(self, input, orig_pcd, return_lattice=False):
        H = self.heads
        key_values = self.keys_values_pred(input)
        keys_res, values = self._norm_keys_values(key_values)
        keys, lattice, kstats = self._lattice(orig_pcd, keys_res)
        z = self.splat.forward_keys(lattice, values)
        occ = self._occupancy(z, keys.size(0))
        with torch.no_grad():
            stats = (occ, kstats[0], kstats[1], None)       # mean / variance of the keys, reduced by the lattice kernel
        result = z
        if return_lattice:
            result = result, lattice
        return result, stats


class MultiHeadAdaIn(_MHCTCore):
    """Style-conditioned MHCT: AdaIN instead of BatchNorm, learnable scalar `scale`
    (init 0) on the key residual, no pts_padding."""

    def __init__(self, model_dim, in_feature_dim, out_model_dim, tensor_size, tensor_dim, heads,
                 n_latent=256, unet=False, scales=False):
        super().__init__()
        self.out_model_dim = out_model_dim
        self.num_latent = n_latent
        self._build_core(model_dim, in_feature_dim, tensor_size, tensor_dim, heads, True)
        self.values_bn = nn.Sequential(AdaIn1dUpd(heads * in_feature_dim, num_latent=n_latent))
        self.keys_bn = nn.Sequential(AdaIn1dUpd(heads * 3, num_latent=n_latent))
        self._build_grid_modules(True)
        self.conv = _grouped_conv(tensor_dim, heads * in_feature_dim, heads)
        self.after = nn.Sequential(AdaIn1dUpd(heads * in_feature_dim, num_latent=n_latent), nn.ReLU(inplace=True))
        self.scale = nn.Parameter(torch.tensor(0, dtype=torch.float32))
        self._build_transform(scales)

    def _forward_pre(self, input, style, orig_pcd, kv=None):
        """Everything up to (not including) `after`; `kv` = (keys_res, values) when the union block computed them."""
        H = self.heads
        if kv is None:
            key_values = forward_style(self.keys_values_pred, input, style)
            k_part, v_part = torch.split(key_values, [H * 3, key_values.size(1) - H * 3], dim=1)   # backward: one cat
            keys_res = forward_style(self.keys_bn, k_part, style)
            values = forward_style(self.values_bn, v_part, style)
        else:
            keys_res, values = kv
        keys, lattice, kstats = self._lattice(orig_pcd, keys_res, self.scale)
        pre, occ = self._core(lattice, values)
        with torch.no_grad():
            # the reference moves these to the host and copies ALL keys to numpy on every
            # forward (multihead_ct_adain.py:127-131); here they stay on the device (no sync)
            stats = (occ, kstats[0], kstats[1], keys.detach())
        return pre, stats, lattice

    def forward(self, input, style, orig_pcd, return_lattice=False):
        pre, stats, lattice = self._forward_pre(input, style, orig_pcd)
        result = forward_style(self.after, pre, style)
        if return_lattice:
            result = result, lattice
        return result, stats


# "auto" (default): only while a HIP graph is being captured — a replayed graph gains 3-4 % from the overlap, eager launches pay
# for the stream switches on the host (the eager classifier step is 14 % slower with them); "1": always; "0": never
SKIP_IN_DGRAD = os.environ.get("CLOUDCT_SKIP_IN_DGRAD", "1") != "0"      # "0": autograd sums the two cotangents of a block's input (A/B)
HEAD_STREAMS = {"0": False, "1": True}.get(os.environ.get("CLOUDCT_HEAD_STREAMS", "auto"), "auto")
_side_streams = {}


def _run_heads(calls, ref):
    """[f() for f in calls] — the per-head chains of a union block (lattice, Splat, grouped conv, Slice of each head: independent
    until the concatenation).  When enabled (HEAD_STREAMS above) head i > 0 runs on its own side stream, forked
    from the current stream and joined before the results are used: the heads' kernels are many and small (a 3D head's raster
    kernels launch 128 workgroups on 256 CUs), so two chains side by side fill the chip and hide each other's launch gaps.
    Autograd replays every op's backward on the stream of its forward, so the backward chains overlap the same way; a HIP-graph
    capture records the fork and join as graph dependencies."""
    on = HEAD_STREAMS if HEAD_STREAMS != "auto" else (ref.is_cuda and torch.cuda.is_current_stream_capturing())
    if not (on and len(calls) > 1 and ref.is_cuda):
        return [f() for f in calls]
    cur = torch.cuda.current_stream(ref.device)
    key = (ref.device.index, len(calls))
    sides = _side_streams.get(key)
    if sides is None:
        sides = _side_streams[key] = [torch.cuda.Stream(device=ref.device) for _ in range(len(calls) - 1)]
        ops.forked_streams.update(s.cuda_stream for s in sides)        # nothing running on them forks again (ops.pw_backward)
    outs = [None] * len(calls)
    for i, side in enumerate(sides, start=1):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            outs[i] = calls[i]()
    outs[0] = calls[0]()
    for i, side in enumerate(sides, start=1):
        cur.wait_stream(side)
        _record(outs[i], cur)
    return outs


def _record(obj, stream):
    """Tell the caching allocator that tensors made on a side stream are used on `stream` from here on."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _record(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record(o, stream)


class _UnionBase(nn.Module):
    """K MHCT blocks on the same input, concatenated on channels, projected back to
    the model width and added to a (possibly projected) residual."""

    def _check(self, features_dims, tensor_sizes, tensor_dims, heads):
        assert len(features_dims) == len(tensor_sizes)
        assert len(features_dims) == len(tensor_dims)
        assert len(features_dims) == len(heads)

    def _common(self, model_dim, features_dims, tensor_sizes, tensor_dims, heads, model_dim_out):
        self._check(features_dims, tensor_sizes, tensor_dims, heads)
        self.model_dim = model_dim
        self.features_dims = features_dims
        self.tensor_sizes = tensor_sizes
        self.tensor_dims = tensor_dims
        self.heads = heads
        self.model_dim_out = model_dim if model_dim_out is None else model_dim_out
        self.prenorm = nn.Sequential()
        return sum(h * f for h, f in zip(heads, features_dims))


class MultiHeadUnion(_UnionBase):
    def __init__(self, model_dim, features_dims, tensor_sizes, tensor_dims, heads, model_dim_out=None, scales=False):
        super().__init__()
        cat_dim = self._common(model_dim, features_dims, tensor_sizes, tensor_dims, heads, model_dim_out)
        self.after = nn.Sequential(
            PointwiseConv1d(cat_dim, self.model_dim_out, kernel_size=1, stride=1, padding=0, bias=False),
            nn.BatchNorm1d(self.model_dim_out),
            nn.ReLU(inplace=True))
        self.shortcut = nn.Sequential()
        if self.model_dim != self.model_dim_out:
            self.shortcut.add_module("shortcut_conv", PointwiseConv1d(self.model_dim, self.model_dim_out, kernel_size=1,
                                                                stride=1, padding=0, bias=False))
            self.shortcut.add_module("shortcut_bn", nn.BatchNorm1d(self.model_dim_out))
        self.attentions = nn.ModuleList([
            MultiHead(model_dim=self.model_dim, in_feature_dim=f, out_model_dim=self.model_dim_out,
                      tensor_size=w, tensor_dim=d, heads=h, scales=scales)
            for f, w, d, h in zip(features_dims, tensor_sizes, tensor_dims, heads)])

    def forward(self, x, orig_pcd):
        x = self.prenorm(x)
        pres, stats = [], []
        # the heads share x: one stacked GEMM for their keys_values_pred projections (and for its gradients)
        convs = [a.keys_values_pred[0] for a in self.attentions]
        kbs, vbs = [a.key_bn for a in self.attentions], [a.values_bn for a in self.attentions]
        if ops.union_keys_values_eligible(x, convs, kbs, vbs):
            if SKIP_IN_DGRAD and len(self.shortcut) == 0 and x.requires_grad and torch.is_grad_enabled():
                # identity shortcut: x leaves the projections' node as an output too, so the shortcut's cotangent is added in the
                # data gradient's epilogue (ops.UnionKeysValuesFn) — autograd's own sum was a 31 us pass per block on the
                # segmenter's critical path (profiles/r6_skip_in_dgrad.txt)
                residual, kvs = ops.union_keys_values(x, convs, kbs, vbs, passthrough=True)
            else:
                residual, kvs = self.shortcut(x), ops.union_keys_values(x, convs, kbs, vbs)
        else:
            residual, kvs = self.shortcut(x), None
        for r, s, _ in _run_heads([(lambda a=a, i=i: a._forward_pre(x, orig_pcd, None if kvs is None else kvs[i]))
                                   for i, a in enumerate(self.attentions)], x):
            pres.append(r)
            stats.append(s)
        # the heads' BatchNorm + ReLU write straight into their channel ranges of the concatenation when they qualify
        norms = [a.after for a in self.attentions]
        if (len(pres) > 1 and all(len(n) == 2 and type(n[1]) is nn.ReLU and ops.bn_relu_eligible(n[0], p)
                                  for n, p in zip(norms, pres)) and ops.norms_share_group([n[0] for n in norms])):
            joined = ops.join_bn_relu(pres, [n[0] for n in norms])
        else:
            joined = torch.cat([run_after(n, p) for n, p in zip(norms, pres)], dim=1)
        return run_after(self.after, joined, residual), stats


class MultiHeadUnionAdaIn(_UnionBase):
    def __init__(self, model_dim, features_dims, tensor_sizes, tensor_dims, heads, model_dim_out=None,
                 n_latent=256, unet=False, scales=False):
        super().__init__()
        cat_dim = self._common(model_dim, features_dims, tensor_sizes, tensor_dims, heads, model_dim_out)
        self.after = nn.Sequential(
            PointwiseConv1d(cat_dim, self.model_dim_out, kernel_size=1, stride=1, padding=0, bias=False),
            AdaIn1dUpd(self.model_dim_out, num_latent=n_latent),
            nn.ReLU(inplace=True))
        self.shortcut = nn.Sequential()
        if self.model_dim != self.model_dim_out:
            self.shortcut.add_module("shortcut_conv", PointwiseConv1d(self.model_dim, self.model_dim_out, kernel_size=1,
                                                                stride=1, padding=0, bias=False))
            self.shortcut.add_module("shortcut_bn", AdaIn1dUpd(self.model_dim_out, num_latent=n_latent))
        self.attentions = nn.ModuleList([
            MultiHeadAdaIn(model_dim=self.model_dim, in_feature_dim=f, out_model_dim=self.model_dim_out,
                           tensor_size=w, tensor_dim=d, n_latent=n_latent, heads=h, unet=unet, scales=scales)
            for f, w, d, h in zip(features_dims, tensor_sizes, tensor_dims, heads)])

    def forward(self, x, style, orig_pcd):
        norms = self._style_norms(style)
        try:
            return self._forward(x, style, orig_pcd)
        finally:
            for m in norms:
                m._gb = None

    def _style_norms(self, style):
        """Project the style vector for every AdaIn1dUpd of the block at once (ops.StyleProjFn) and hand each norm its slice for
        the span of this forward; returns the norms to reset."""
        if not (style.is_cuda and style.dtype == torch.float32 and style.dim() == 2):
            return []
        mods = self.__dict__.get("_adain_norms")
        if mods is None:
            mods = [m for m in self.modules() if type(m) is AdaIn1dUpd]
            self.__dict__["_adain_norms"] = mods
        mods = [m for m in mods if m.linear.in_features == style.size(1) and m.linear.bias is not None]
        if len(mods) < 2:
            return []
        args = []
        for m in mods:
            args += [m.linear.weight, m.linear.bias]
        for m, gb in zip(mods, ops.StyleProjFn.apply(style, *args)):
            m._gb = gb
        return mods

    def _forward(self, x, style, orig_pcd):
        x = self.prenorm(x)
        pres, stats = [], []
        # (identity shortcut: its cotangent rides the stacked projections' data gradient, see MultiHeadUnion.forward)
        skip = SKIP_IN_DGRAD and len(self.shortcut) == 0 and x.requires_grad and torch.is_grad_enabled()
        kvs = self._fused_keys_values(x, style, passthrough=skip)
        if skip and kvs is not None:
            residual, kvs = kvs
        else:
            residual = forward_style(self.shortcut, x, style)
        for r, s, _ in _run_heads([(lambda a=a, i=i: a._forward_pre(x, style, orig_pcd, None if kvs is None else kvs[i]))
                                   for i, a in enumerate(self.attentions)], x):
            pres.append(r)
            stats.append(s)
        afters = [a.after for a in self.attentions]
        fusable = (len(pres) > 1 and all(len(n) == 2 and type(n[0]) is AdaIn1dUpd and type(n[1]) is nn.ReLU for n in afters)
                   and all(p.is_cuda and p.dtype == torch.float32 and p.dim() == 3 for p in pres))
        if fusable:        # the heads' AdaIN + ReLU write straight into their channel ranges of the concatenation
            args = []
            for n, p in zip(afters, pres):
                args += [p, n[0].gamma_beta(style)]
            joined = ops.JoinAdaInReluFn.apply(len(pres), afters[0][0].instance_norm.eps, *args)
        else:
            joined = torch.cat([forward_style(n, p, style) for n, p in zip(afters, pres)], dim=1)
        return forward_style(self.after, joined, style, residual), stats

    def _fused_keys_values(self, x, style, passthrough=False):
        """One stacked GEMM for all heads' keys_values_pred + their AdaIN norms (ops.UnionKeysValuesAdaInFn), or None; with
        passthrough: (x as an output of that node, the list)."""
        atts = list(self.attentions)
        if not (len(atts) > 1 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()):
            return None
        args, eps = [], None
        for a in atts:
            kvp, kb, vb = a.keys_values_pred, a.keys_bn, a.values_bn
            if not (len(kvp) == 1 and isinstance(kvp[0], nn.Conv1d) and kvp[0].kernel_size == (1,) and kvp[0].bias is None
                    and kvp[0].stride == (1,) and kvp[0].padding == (0,) and kvp[0].groups == 1 and kvp[0].dilation == (1,)
                    and kvp[0].in_channels == x.size(1)
                    and len(kb) == 1 and type(kb[0]) is AdaIn1dUpd and len(vb) == 1 and type(vb[0]) is AdaIn1dUpd
                    and kvp[0].out_channels == kb[0].num_features + vb[0].num_features):
                return None
            e = kb[0].instance_norm.eps
            if vb[0].instance_norm.eps != e or (eps is not None and eps != e):
                return None
            eps = e
            args += [kvp[0].weight, kb[0].gamma_beta(style), vb[0].gamma_beta(style)]
        outs = ops.UnionKeysValuesAdaInFn.apply(-len(atts) if passthrough else len(atts), x, eps, *args)
        if passthrough:
            return outs[0], [(outs[1 + 2 * i], outs[2 + 2 * i]) for i in range(len(atts))]
        return [(outs[2 * i], outs[2 * i + 1]) for i in range(len(atts))]

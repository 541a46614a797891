"""CPU oracle for the Multi-Headed Cloud Transform hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`cloud_transformers_amd/`) may import this module: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
there only as the checker / the reported CPU baseline.

It is a plain-PyTorch (CPU, fp32) restatement of the reference's algorithm,
op for op, including the materialised `(B,H,C,V,N)` intermediates the
reference builds, so that timing it is a fair "port" of the reference's CPU
path.  Each function cites the reference lines it follows.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md §4).  This oracle is pinned against outputs of the reference ITSELF,
imported in the build container by `tests/golden/gen_golden.py` (with stand-ins
for the two third-party functions `torch_scatter.scatter_max` and
`pytorch3d...so3_exponential_map`, which are not vendored in the reference);
`tests/test_oracle_golden.py` checks every function below against those
fixtures (indices exact, floats to <=1e-6).
"""
import math

import torch


def _sizes(tensor_size, dim):
    """layers/cloud_transform.py:41-46 — int -> [W]*dim, tuple kept."""
    if isinstance(tensor_size, int):
        return [tensor_size] * dim
    sizes = [int(w) for w in tensor_size]
    assert len(sizes) == dim
    return sizes


class _Balance(torch.autograd.Function):
    """layers/cloud_transform.py:12-26 — forward x*scale, backward passes the
    cotangent through UNSCALED."""

    @staticmethod
    def forward(ctx, x, scale):
        return x * scale

    @staticmethod
    def backward(ctx, g):
        return g, None


def positions(keys, tensor_size, heads, dim):
    """layers/cloud_transform.py:72-121 + layers/utils.py:100-186.

    keys f32[B, H*dim, N] -> (local_coordinate f32[B,H,V,N], flattened_index i64[B,H,V,N]).
    """
    W = _sizes(tensor_size, dim)
    B, HD, N = keys.shape
    assert HD == heads * dim
    eps = 1e-7
    k = keys.reshape(B * heads, dim, N).clamp(-1 + eps, 1 - eps)
    mod = torch.tensor(W, dtype=torch.float32)[None, :, None]
    s = _Balance.apply(k + 1.0, (mod - 1) * 0.5)          # cloud_transform.py:94
    f = s.floor()
    w0 = (f + 1) - s                                       # weight of the low corner, per axis
    w1 = s - f                                             # weight of the high corner
    V = 1 << dim
    ws, cells = [], []
    for v in range(V):
        off = [(v >> j) & 1 for j in range(dim)]           # v = dx + 2dy (+ 4dz)
        wv = None
        for j in range(dim):
            wj = w1[:, j] if off[j] else w0[:, j]
            wv = wj if wv is None else wv * wj             # ((wx*wy)*wz): utils.py:139-146
        ws.append(wv)
        c = [f[:, j].long() + off[j] for j in range(dim)]
        if dim == 3:                                       # cloud_transform.py:113-119
            cells.append(c[0] * W[1] * W[2] + c[1] * W[2] + c[2])
        else:
            cells.append(c[0] * W[1] + c[1])
    lc = torch.stack(ws, dim=1).reshape(B, heads, V, N)
    idx = torch.stack(cells, dim=1).reshape(B, heads, V, N)
    return lc, idx


def splat(lc, idx, features, pts_padding, tensor_size, heads, dim, reduce="max"):
    """layers/cloud_transform.py:131-180.

    reduce="max": what the reference executes — torch_scatter.scatter_max into a
    ZERO-initialised out (:164-173), i.e. z = max(0, max contributions).
    reduce="sum": scatter_add_, the variant BASELINE.json's north_star names.
    """
    W = _sizes(tensor_size, dim)
    B, HC, N = features.shape
    assert features.dtype == torch.float32 and HC % heads == 0
    C = HC // heads
    G = math.prod(W)
    f = features.reshape(B, heads, C, N)
    if pts_padding is not None:
        f = f * pts_padding[:, None, None, :]
    pre = f[:, :, :, None] * lc[:, :, None]                # (B,H,C,V,N) materialised, :161
    z0 = torch.zeros(B, heads, C, G, dtype=torch.float32)
    index = idx[:, :, None].reshape(B, heads, 1, -1).expand(B, heads, C, -1)
    src = pre.reshape(B, heads, C, -1)
    if reduce == "max":
        z = z0.scatter_reduce(3, index, src, reduce="amax", include_self=True)
    elif reduce == "sum":
        z = z0.scatter_add(3, index, src)
    else:
        raise ValueError(reduce)
    return z.reshape(B, heads * C, *W)


def slice_(lc, idx, grid, pts_padding, tensor_size, heads, dim):
    """layers/cloud_transform.py:190-227 (gather with the index expanded over C)."""
    W = _sizes(tensor_size, dim)
    B, H, V, N = lc.shape
    assert grid.shape[1] % heads == 0
    C = grid.shape[1] // heads
    index = idx[:, :, None].expand(-1, -1, C, -1, -1).reshape(B, heads, C, -1)
    g = torch.gather(grid.reshape(B, heads, C, -1), 3, index).reshape(B, heads, C, V, N)
    out = (g * lc[:, :, None]).sum(dim=3).reshape(B, heads * C, N)
    if pts_padding is not None:
        out = out * pts_padding[:, None, :]
    return out


def splat_slice_step(keys, feat, cot, tensor_size, heads, dim, reduce="max"):
    """One fwd+bwd pass of positions -> Splat -> Slice (the bench 'step'),
    returning (out, g_feat, g_keys).  Used as the cpu_baseline 'port'."""
    keys = keys.detach().clone().requires_grad_(True)
    feat = feat.detach().clone().requires_grad_(True)
    lc, idx = positions(keys, tensor_size, heads, dim)
    z = splat(lc, idx, feat, None, tensor_size, heads, dim, reduce)
    out = slice_(lc, idx, z, None, tensor_size, heads, dim)
    out.backward(cot)
    return out.detach(), feat.grad, keys.grad


# ---------------------------------------------------------------------------
# per-head rigid transform (layers/utils.py:9-61) with the Rodrigues map that
# pytorch3d.transforms.so3.so3_exponential_map publishes (eps = 1e-4 clamp on
# the squared norm)
# ---------------------------------------------------------------------------
def so3_exp(log_R, eps=1e-4):
    n2 = (log_R * log_R).sum(1)
    th = torch.clamp(n2, eps).sqrt()
    a = th.sin() / th
    b = (1.0 - th.cos()) / (th * th)
    x, y, z = log_R.unbind(1)
    o = torch.zeros_like(x)
    K = torch.stack([o, -z, y, z, o, -x, -y, x, o], 1).reshape(-1, 3, 3)
    return a[:, None, None] * K + b[:, None, None] * (K @ K) + torch.eye(3)[None]


def rigid_transform(pcd, log_R, shift, scales, dim):
    """pcd f32[B,H,3,N] -> [B,H,dim,N]; row-vector times R (utils.py:27-29,54-56);
    planes keep the first two rotated coordinates (utils.py:58-61)."""
    p = pcd + shift[None, :, :, None]
    p = torch.einsum("bhcp,hcn->bhnp", p, so3_exp(log_R))
    if dim == 2:
        p = p[:, :, :2]
    if scales is not None:
        p = p * scales[None, :, :, None]
    return p


# ---------------------------------------------------------------------------
# MHCT blocks, functional form over a reference-layout state dict
# (layers/multihead_ct.py:82-118,182-198; multihead_ct_pool.py:55-86;
#  multihead_ct_adain.py:104-136,202-218; layers/utils.py:82-97)
# ---------------------------------------------------------------------------
def _bn(x, sd, prefix, train, eps=1e-5):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if train:
        m = x.mean(dim=(0, 2))
        v = x.var(dim=(0, 2), unbiased=False)
    else:
        m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - m[None, :, None]) / torch.sqrt(v[None, :, None] + eps) * w[None, :, None] + b[None, :, None]


def _adain(x, style, sd, prefix, eps=1e-5):
    m = x.mean(dim=2, keepdim=True)
    v = x.var(dim=2, unbiased=False, keepdim=True)
    xn = (x - m) / torch.sqrt(v + eps)
    vb = torch.nn.functional.linear(style, sd[prefix + ".linear.weight"], sd[prefix + ".linear.bias"])
    vb = vb.reshape(-1, 2, x.shape[1])
    return xn * (vb[:, 0][:, :, None] + 1) + vb[:, 1][:, :, None]


def _gconv(z, sd, prefix, heads, dim):
    fn = torch.nn.functional.conv3d if dim == 3 else torch.nn.functional.conv2d
    return fn(z, sd[prefix + ".weight"], sd[prefix + ".bias"], stride=1, padding=1, groups=heads)


def multihead(sd, x, orig_pcd, *, in_feature_dim, tensor_size, tensor_dim, heads,
              train=False, pad=None, pool=False, style=None, prefix=""):
    """Returns (result, occ, mean(keys), var(keys)).  style!=None selects the
    AdaIN flavour; pool=True stops after Splat."""
    H, C, d = heads, in_feature_dim, tensor_dim
    B, _, N = x.shape
    p = prefix
    kv = torch.nn.functional.conv1d(x, sd[p + "keys_values_pred.0.weight"])
    if style is None:
        keys_res = _bn(kv[:, :3 * H], sd, p + "key_bn", train)
        values = _bn(kv[:, 3 * H:], sd, p + "values_bn", train)
        kscale = 1.0
    else:
        keys_res = _adain(kv[:, :3 * H], style, sd, p + "keys_bn.0")
        values = _adain(kv[:, 3 * H:], style, sd, p + "values_bn.0")
        kscale = sd[p + "scale"]
    scales = sd.get(p + "transform.scales")
    keys = rigid_transform(orig_pcd[:, None] + kscale * keys_res.reshape(B, H, 3, N),
                           sd[p + "transform.log_R"], sd[p + "transform.shift"], scales, d)
    keys = keys.reshape(B, H * d, N)
    lattice = torch.tanh(keys)
    lc, idx = positions(lattice, tensor_size, H, d)
    z = splat(lc, idx, values, pad, tensor_size, H, d)
    occ = (z.abs() > 1e-9).sum().float() / (B * C * H)
    if pool:
        return z, occ, keys.mean(), keys.var()
    s = slice_(lc, idx, _gconv(z, sd, p + "conv.0", H, d), pad, tensor_size, H, d)
    if style is None:
        res = torch.relu(_bn(s, sd, p + "after.0", train))
    else:
        res = torch.relu(_adain(s, style, sd, p + "after.0"))
    return res, occ, keys.mean(), keys.var()


def multihead_union(sd, x, orig_pcd, *, features_dims, tensor_sizes, tensor_dims, heads,
                    train=False, style=None):
    outs, occs = [], []
    for i, (C, W, d, H) in enumerate(zip(features_dims, tensor_sizes, tensor_dims, heads)):
        r, occ, _, _ = multihead(sd, x, orig_pcd, in_feature_dim=C, tensor_size=W, tensor_dim=d,
                                 heads=H, train=train, style=style, prefix=f"attentions.{i}.")
        outs.append(r)
        occs.append(occ)
    cat = torch.cat(outs, dim=1)
    y = torch.nn.functional.conv1d(cat, sd["after.0.weight"])
    if style is None:
        y = torch.relu(_bn(y, sd, "after.1", train))
    else:
        y = torch.relu(_adain(y, style, sd, "after.1"))
    if "shortcut.shortcut_conv.weight" in sd:
        r = torch.nn.functional.conv1d(x, sd["shortcut.shortcut_conv.weight"])
        if style is None:
            r = _bn(r, sd, "shortcut.shortcut_bn", train)
        else:
            r = _adain(r, style, sd, "shortcut.shortcut_bn")
    else:
        r = x
    return r + y, occs


# ---------------------------------------------------------------------------
# Chamfer (chamfer_extension/chamfer.cu:12-134 forward, :155-174 backward)
# ---------------------------------------------------------------------------
def chamfer_fwd(xyz1, xyz2):
    """dist1[b,i] = min_j |x1_i - x2_j|^2 (argmin = lowest j on ties: strict '<'
    at chamfer.cu:36,46), and symmetrically for cloud 2."""
    d = ((xyz1[:, :, None, :] - xyz2[:, None, :, :]) ** 2).sum(-1)   # (B,n,m)
    d1, i1 = d.min(dim=2)
    d2, i2 = d.min(dim=1)
    return d1, d2, i1.int(), i2.int()


def chamfer_bwd(xyz1, xyz2, g1, g2, i1, i2):
    """chamfer.cu:155-174: g=2*grad_dist; grad_xyz1[i] += g*(x1_i-x2_idx);
    grad_xyz2[idx] -= g*(x1_i-x2_idx); and the same with the clouds swapped."""
    B = xyz1.shape[0]
    ga, gb = torch.zeros_like(xyz1), torch.zeros_like(xyz2)
    for b in range(B):
        nb = xyz2[b][i1[b].long()]
        t = 2 * g1[b][:, None] * (xyz1[b] - nb)
        ga[b] += t
        gb[b].index_add_(0, i1[b].long(), -t)
        na = xyz1[b][i2[b].long()]
        t = 2 * g2[b][:, None] * (xyz2[b] - na)
        gb[b] += t
        ga[b].index_add_(0, i2[b].long(), -t)
    return ga, gb


def fscore(gt, pr, th=0.01):
    """(fscore, precision, recall) of utils/f1_metric.py:9-30 for one pair of clouds gt [n,3], pr [m,3].
    Nearest-neighbour distances come from a float64 KD-tree (scipy), as open3d's
    compute_point_cloud_distance does in the reference (:13-14): d1 = gt->pr, d2 = pr->gt,
    recall = |d2 < th| / |d2|, precision = |d1 < th| / |d1| (:17-18)."""
    import numpy as np
    from scipy.spatial import cKDTree
    gt = np.asarray(gt, dtype=np.float64)
    pr = np.asarray(pr, dtype=np.float64)
    if len(gt) == 0 or len(pr) == 0:
        return 0.0, 0.0, 0.0
    d1 = cKDTree(pr).query(gt)[0]
    d2 = cKDTree(gt).query(pr)[0]
    recall = float((d2 < th).sum()) / len(d2)
    precision = float((d1 < th).sum()) / len(d1)
    f = 2 * recall * precision / (recall + precision) if recall + precision > 0 else 0.0
    return f, precision, recall

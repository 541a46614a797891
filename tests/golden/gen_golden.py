#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference). Nothing from the
reference's source travels: only inputs and the outputs it produced are saved.

The reference imports two third-party modules that are not installed here and
whose source is not under /root/reference (install_deps.sh:6,10):

  * torch_scatter.scatter_max    (call site layers/cloud_transform.py:171-173)
  * pytorch3d.transforms.so3.so3_exponential_map (call sites layers/utils.py:29,56)

They are replaced by the two stand-ins below, which restate the PUBLISHED
semantics of those functions (torch-scatter docs: "maximum of all values from
src at the indices given by index, starting from out"; pytorch3d docs:
Rodrigues formula with the squared-norm clamp eps=1e-4). The stand-in's
backward differs from torch_scatter's only on exact ties, so every golden
gradient below is produced from tie-free inputs (continuous random values).

usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# stand-ins for the two absent third-party modules
# --------------------------------------------------------------------------
def _scatter_max(src, index, dim=-1, out=None, dim_size=None):
    assert out is not None
    res = out.scatter_reduce(dim, index.expand_as(src), src, reduce="amax",
                             include_self=True)
    return res, None


def _so3_exponential_map(log_rot, eps=1e-4):
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, eps).sqrt()
    rot_angles_inv = 1.0 / rot_angles
    fac1 = rot_angles_inv * rot_angles.sin()
    fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
    x, y, z = log_rot[:, 0], log_rot[:, 1], log_rot[:, 2]
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], dim=1).reshape(-1, 3, 3)
    KK = torch.bmm(K, K)
    return (fac1[:, None, None] * K + fac2[:, None, None] * KK
            + torch.eye(3, dtype=log_rot.dtype)[None])


def _install_shims():
    ts = types.ModuleType("torch_scatter")
    ts.scatter_max = _scatter_max
    sys.modules["torch_scatter"] = ts
    p3 = types.ModuleType("pytorch3d")
    p3t = types.ModuleType("pytorch3d.transforms")
    p3s = types.ModuleType("pytorch3d.transforms.so3")
    p3s.so3_exponential_map = _so3_exponential_map
    p3.transforms = p3t
    p3t.so3 = p3s
    sys.modules["pytorch3d"] = p3
    sys.modules["pytorch3d.transforms"] = p3t
    sys.modules["pytorch3d.transforms.so3"] = p3s
    sys.path.insert(0, REF)


def _np(t):
    return t.detach().cpu().numpy().copy()


# --------------------------------------------------------------------------
# 1. positions
# --------------------------------------------------------------------------
def gen_positions(ct):
    cases = {}
    specs = [("2d_w8", 2, 8), ("2d_w32", 2, 32), ("2d_w128", 2, 128),
             ("2d_w16x24", 2, (16, 24)), ("3d_w8", 3, 8), ("3d_w16", 3, 16),
             ("3d_w32", 3, 32), ("3d_w3x4x5", 3, (3, 4, 5))]
    g = torch.Generator().manual_seed(101)
    for name, dim, W in specs:
        B, H, N = 2, 3, 96
        keys = torch.tanh(torch.randn(B, H * dim, N, generator=g) * 1.5)
        # edge values in the first points of every row
        edge = torch.tensor([-1.0, 1.0, -0.99999994, 0.99999994, 0.0, -0.5, 0.5,
                             -1.5, 1.5, 1e-8, -1e-8, 0.99999988, -0.99999988,
                             0.25, -0.75, 0.9375], dtype=torch.float32)
        keys[:, :, :edge.numel()] = edge
        # exact cell-edge values for integer W: s = (k+1)*(W-1)/2 integer
        Wl = [W] * dim if isinstance(W, int) else list(W)
        for j in range(dim):
            Wm = Wl[j] - 1
            ks = torch.arange(0, Wm + 1, dtype=torch.float32) * (2.0 / Wm) - 1.0
            n0 = edge.numel()
            m = min(ks.numel(), N - n0)
            keys[:, j::dim, n0:n0 + m] = ks[:m]
        keys = keys.clone().requires_grad_(True)
        mod = ct.DifferentiablePositions(tensor_size=W, heads=H, dim=dim)
        lc, idx = mod(keys)
        cot = torch.randn(lc.shape, generator=g)
        (lc * cot).sum().backward()
        cases[name] = dict(dim=dim, W=np.array(Wl, dtype=np.int32), H=H,
                           keys=_np(keys), lc=_np(lc), idx=_np(idx),
                           cot_lc=_np(cot), g_keys=_np(keys.grad))
    np.savez_compressed(os.path.join(OUT, "positions.npz"),
                        **{f"{c}/{k}": v for c, d in cases.items() for k, v in d.items()})
    print("positions:", list(cases))


# --------------------------------------------------------------------------
# 2. splat / slice forward + backward
# --------------------------------------------------------------------------
def gen_splat_slice(ct):
    specs = [
        # name, B, H, C, N, dim, W, pad
        ("cfg0_2d_w32", 2, 4, 8, 1024, 2, 32, None),       # BASELINE configs[0] shape
        ("2d_w32_pad", 1, 2, 4, 512, 2, 32, "float"),
        ("3d_w8", 2, 2, 4, 512, 3, 8, None),
        ("3d_w8_padint", 2, 2, 4, 512, 3, 8, "int32"),
        ("2d_w16x24", 1, 3, 5, 300, 2, (16, 24), None),
        ("3d_w3x4x5", 2, 2, 3, 77, 3, (3, 4, 5), "float"),
        ("2d_w8_c1", 1, 1, 1, 64, 2, 8, None),
        ("2d_w32_c16", 1, 2, 16, 777, 2, 32, None),
        ("3d_w16_c16", 1, 1, 16, 640, 3, 16, None),
    ]
    g = torch.Generator().manual_seed(202)
    out = {}
    for name, B, H, C, N, dim, W, pad in specs:
        Wl = [W] * dim if isinstance(W, int) else list(W)
        G = int(np.prod(Wl))
        pos = ct.DifferentiablePositions(tensor_size=W, heads=H, dim=dim)
        splat = ct.Splat(tensor_size=W, heads=H, dim=dim)
        slc = ct.Slice(tensor_size=W, heads=H, dim=dim)
        keys0 = torch.tanh(torch.randn(B, H * dim, N, generator=g))
        feat0 = torch.randn(B, H * C, N, generator=g)
        grid0 = torch.randn(B, H * C, *Wl, generator=g)
        cot_z = torch.randn(B, H * C, *Wl, generator=g)
        cot_o = torch.randn(B, H * C, N, generator=g)
        padt = None
        if pad is not None:
            padt = (torch.rand(B, N, generator=g) > 0.25)
            padt = padt.float() if pad == "float" else padt.int()
        d = dict(dim=dim, W=np.array(Wl, dtype=np.int32), H=H, C=C,
                 keys=_np(keys0), feat=_np(feat0), grid=_np(grid0),
                 cot_z=_np(cot_z), cot_o=_np(cot_o))
        if padt is not None:
            d["pad"] = _np(padt)

        # (a) Splat alone
        keys = keys0.clone().requires_grad_(True)
        feat = feat0.clone().requires_grad_(True)
        lc, idx = pos(keys)
        z = splat(lc, idx, feat, padt)
        (z * cot_z).sum().backward()
        d.update(z=_np(z), splat_g_feat=_np(feat.grad), splat_g_keys=_np(keys.grad))

        # (b) Slice alone
        keys = keys0.clone().requires_grad_(True)
        grid = grid0.clone().requires_grad_(True)
        lc, idx = pos(keys)
        o = slc(lc, idx, grid, padt)
        (o * cot_o).sum().backward()
        d.update(sliced=_np(o), slice_g_grid=_np(grid.grad), slice_g_keys=_np(keys.grad))

        # (c) chain Slice(Splat(.)) — the bench path
        keys = keys0.clone().requires_grad_(True)
        feat = feat0.clone().requires_grad_(True)
        lc, idx = pos(keys)
        o = slc(lc, idx, splat(lc, idx, feat, padt), padt)
        (o * cot_o).sum().backward()
        d.update(chain_out=_np(o), chain_g_feat=_np(feat.grad), chain_g_keys=_np(keys.grad))
        out[name] = d
    np.savez_compressed(os.path.join(OUT, "splat_slice.npz"),
                        **{f"{c}/{k}": v for c, d in out.items() for k, v in d.items()})
    print("splat_slice:", list(out))


# --------------------------------------------------------------------------
# 3. so3 exponential map + rigid transforms (layers/utils.py:9-61)
# --------------------------------------------------------------------------
def gen_transforms(lu):
    g = torch.Generator().manual_seed(303)
    H = 6
    out = {}
    for name, cls, scales in [("plane", lu.PlaneTransformer, False),
                              ("plane_scales", lu.PlaneTransformer, True),
                              ("vol", lu.VolTransformer, False),
                              ("vol_scales", lu.VolTransformer, True)]:
        m = cls(H, scales=scales)
        with torch.no_grad():
            m.log_R.copy_(torch.randn(H, 3, generator=g))
            m.log_R[0].mul_(1e-4)      # near-zero rotation (eps clamp branch)
            m.shift.copy_(torch.randn(H, 3, generator=g) * 0.1)
            if scales:
                m.scales.copy_(1 + 0.2 * torch.randn(m.scales.shape, generator=g))
        pcd = torch.rand(2, H, 3, 50, generator=g) * 2 - 1
        y = m(pcd)
        d = dict(log_R=_np(m.log_R), shift=_np(m.shift), pcd=_np(pcd), out=_np(y))
        if scales:
            d["scales"] = _np(m.scales)
        out[name] = d
    np.savez_compressed(os.path.join(OUT, "transforms.npz"),
                        **{f"{c}/{k}": v for c, d in out.items() for k, v in d.items()})
    print("transforms:", list(out))


# --------------------------------------------------------------------------
# 4. MHCT blocks (layers/multihead_ct*.py)
# --------------------------------------------------------------------------
def _randomize(mod, g):
    """Non-trivial parameters/buffers so that BN/AdaIN/key paths are exercised."""
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() <= 1 or n.endswith("log_R") or n.endswith("shift") or n.endswith("scales"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / np.sqrt(p[0].numel())))
        for n, b in mod.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))


def _sd(mod):
    return {k: _np(v) for k, v in mod.state_dict().items()}


def gen_blocks():
    from layers import multihead_ct as mh
    from layers import multihead_ct_pool as mhp
    from layers import multihead_ct_adain as mha
    g = torch.Generator().manual_seed(404)
    out = {}

    def run(name, mod, args_fn, train):
        mod.train(train)
        args = args_fn()
        res, stats = mod(*args)
        if isinstance(stats, list):
            occ = np.array([float(s[0]) for s in stats], dtype=np.float32)
            mean = np.array([float(s[1]) for s in stats], dtype=np.float32)
            var = np.array([float(s[2]) for s in stats], dtype=np.float32)
        else:
            occ = np.array([float(stats[0])], dtype=np.float32)
            mean = np.array([float(stats[1])], dtype=np.float32)
            var = np.array([float(stats[2])], dtype=np.float32)
        return dict(out=_np(res), occ=occ, mean=mean, var=var)

    B, N, D = 2, 256, 32
    x = torch.randn(B, D, N, generator=g)
    pcd = torch.rand(B, 3, N, generator=g) * 2 - 1
    style = torch.randn(B, 24, generator=g)

    # MultiHead 2D / 3D, eval + train
    for tag, dim, W, C, H, scales in [("mh2d", 2, 16, 4, 4, False), ("mh3d", 3, 8, 4, 2, True)]:
        m = mh.MultiHead(model_dim=D, in_feature_dim=C, out_model_dim=D, tensor_size=W,
                         tensor_dim=dim, heads=H, scales=scales)
        _randomize(m, g)
        sd = _sd(m)
        d = dict(x=_np(x), pcd=_np(pcd))
        for mode in ("eval", "train"):
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
            r = run(tag, m, lambda: (x, pcd), mode == "train")
            d.update({f"{mode}_{k}": v for k, v in r.items()})
        d.update({f"sd/{k}": v for k, v in sd.items()})
        out[tag] = d

    # MultiHead with padding tuple
    m = mh.MultiHead(model_dim=D, in_feature_dim=4, out_model_dim=D, tensor_size=16,
                     tensor_dim=2, heads=4)
    _randomize(m, g)
    padt = (torch.rand(B, N, generator=g) > 0.3).float()
    m.eval()
    sd = _sd(m)
    res, stats = m(x, (pcd, padt))
    d = dict(x=_np(x), pcd=_np(pcd), pad=_np(padt), eval_out=_np(res),
             eval_occ=np.array([float(stats[0])], dtype=np.float32))
    d.update({f"sd/{k}": v for k, v in sd.items()})
    out["mh2d_pad"] = d

    # MultiHeadUnion (2D + 3D), with a shortcut conv (model_dim_out != model_dim)
    for tag, dout in [("union", None), ("union_proj", 48)]:
        m = mh.MultiHeadUnion(model_dim=D, features_dims=[4, 4], tensor_sizes=[16, 8],
                              tensor_dims=[2, 3], heads=[4, 2], model_dim_out=dout)
        _randomize(m, g)
        sd = _sd(m)
        d = dict(x=_np(x), pcd=_np(pcd))
        for mode in ("eval", "train"):
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
            r = run(tag, m, lambda: (x, pcd), mode == "train")
            d.update({f"{mode}_{k}": v for k, v in r.items()})
        d.update({f"sd/{k}": v for k, v in sd.items()})
        out[tag] = d

    # MultiHeadPool
    m = mhp.MultiHeadPool(model_dim=D, in_feature_dim=4, tensor_size=8, tensor_dim=3, heads=2)
    _randomize(m, g)
    sd = _sd(m)
    d = dict(x=_np(x), pcd=_np(pcd))
    for mode in ("eval", "train"):
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        r = run("pool", m, lambda: (x, pcd), mode == "train")
        d.update({f"{mode}_{k}": v for k, v in r.items()})
    d.update({f"sd/{k}": v for k, v in sd.items()})
    out["pool"] = d

    # AdaIN variants
    m = mha.MultiHeadAdaIn(model_dim=D, in_feature_dim=4, out_model_dim=D, tensor_size=16,
                           tensor_dim=2, heads=4, n_latent=24)
    _randomize(m, g)
    sd = _sd(m)
    m.eval()
    res, stats = m(x, style, pcd)
    d = dict(x=_np(x), pcd=_np(pcd), style=_np(style), eval_out=_np(res),
             eval_occ=np.array([float(stats[0])], dtype=np.float32),
             eval_mean=np.array([float(stats[1])], dtype=np.float32),
             eval_var=np.array([float(stats[2])], dtype=np.float32))
    d.update({f"sd/{k}": v for k, v in sd.items()})
    out["adain"] = d

    m = mha.MultiHeadUnionAdaIn(model_dim=D, features_dims=[4, 4], tensor_sizes=[16, 8],
                                tensor_dims=[2, 3], heads=[4, 2], n_latent=24)
    _randomize(m, g)
    sd = _sd(m)
    m.eval()
    res, stats = m(x, style, pcd)
    d = dict(x=_np(x), pcd=_np(pcd), style=_np(style), eval_out=_np(res),
             eval_occ=np.array([float(s[0]) for s in stats], dtype=np.float32))
    d.update({f"sd/{k}": v for k, v in sd.items()})
    out["union_adain"] = d

    np.savez_compressed(os.path.join(OUT, "blocks.npz"),
                        **{f"{c}/{k}": v for c, d in out.items() for k, v in d.items()})
    print("blocks:", list(out))


# --------------------------------------------------------------------------
# 5. Chamfer: the reference's own pure-torch O(n^2) restatement
#    (chamfer_extension/chamfer_pytorch.py:4-14; n == m only)
# --------------------------------------------------------------------------
def gen_chamfer():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_chamfer_pytorch", os.path.join(REF, "chamfer_extension", "chamfer_pytorch.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    g = torch.Generator().manual_seed(505)
    a = torch.rand(2, 256, 3, generator=g).requires_grad_(True)
    b = torch.rand(2, 256, 3, generator=g).requires_grad_(True)
    res = cp.dist_chamfer(a, b)
    # autograd through the reference's own formulation: the cotangent of a min goes to its arg-min element, so these
    # gradients pin the nearest-neighbour INDICES and the gradient formula of the native kernels (chamfer.cu:155-174)
    cots = [torch.randn(r.shape, generator=g) for r in res]
    sum((r * c).sum() for r, c in zip(res, cots)).backward()
    d = dict(xyz1=_np(a), xyz2=_np(b), g_xyz1=_np(a.grad), g_xyz2=_np(b.grad))
    for i, (r, c) in enumerate(zip(res, cots)):
        d[f"ret{i}"] = _np(r)
        d[f"cot{i}"] = _np(c)
    np.savez_compressed(os.path.join(OUT, "chamfer.npz"), **d)
    print("chamfer: returns", [tuple(r.shape) for r in res])


def main():
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    sys.dont_write_bytecode = True
    torch.set_num_threads(1)
    _install_shims()
    if len(sys.argv) > 1 and sys.argv[1] == "chamfer":      # regenerate this fixture alone
        return gen_chamfer()
    from layers import cloud_transform as ct
    from layers import utils as lu
    gen_positions(ct)
    gen_splat_slice(ct)
    gen_transforms(lu)
    gen_blocks()
    gen_chamfer()


if __name__ == "__main__":
    main()

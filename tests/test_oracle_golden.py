"""Pins the CPU oracle (oracle/ref_cpu.py) against outputs of the reference
itself (tests/golden/*.npz, produced by tests/golden/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.conftest import load_golden

T = torch.from_numpy


def _W(d):
    return tuple(int(w) for w in d["W"])


@pytest.mark.parametrize("case", sorted(load_golden("positions")))
def test_positions(case):
    d = load_golden("positions")[case]
    dim, H, W = int(d["dim"]), int(d["H"]), _W(d)
    keys = T(d["keys"]).requires_grad_(True)
    lc, idx = R.positions(keys, W, H, dim)
    assert idx.dtype == torch.int64
    assert np.array_equal(idx.numpy(), d["idx"])                      # bit-exact indices
    assert np.array_equal(lc.detach().numpy(), d["lc"])               # bit-identical weights
    (lc * T(d["cot_lc"])).sum().backward()
    np.testing.assert_allclose(keys.grad.numpy(), d["g_keys"], rtol=0, atol=1e-6)
    # invariants: weights sum to 1, indices in range
    np.testing.assert_allclose(lc.detach().sum(2).numpy(), 1.0, atol=1e-5)
    assert idx.min() >= 0 and idx.max() < int(np.prod(W))


@pytest.mark.parametrize("case", sorted(load_golden("splat_slice")))
def test_splat_slice(case):
    d = load_golden("splat_slice")[case]
    dim, H, W = int(d["dim"]), int(d["H"]), _W(d)
    pad = T(d["pad"]) if "pad" in d else None

    keys = T(d["keys"]).requires_grad_(True)
    feat = T(d["feat"]).requires_grad_(True)
    lc, idx = R.positions(keys, W, H, dim)
    z = R.splat(lc, idx, feat, pad, W, H, dim)
    assert np.array_equal(z.detach().numpy(), d["z"])                 # max is order-independent
    (z * T(d["cot_z"])).sum().backward()
    np.testing.assert_allclose(feat.grad.numpy(), d["splat_g_feat"], atol=1e-6)
    np.testing.assert_allclose(keys.grad.numpy(), d["splat_g_keys"], atol=2e-6)

    keys = T(d["keys"]).requires_grad_(True)
    grid = T(d["grid"]).requires_grad_(True)
    lc, idx = R.positions(keys, W, H, dim)
    o = R.slice_(lc, idx, grid, pad, W, H, dim)
    np.testing.assert_allclose(o.detach().numpy(), d["sliced"], atol=1e-6)
    (o * T(d["cot_o"])).sum().backward()
    np.testing.assert_allclose(grid.grad.numpy(), d["slice_g_grid"], atol=1e-5)
    np.testing.assert_allclose(keys.grad.numpy(), d["slice_g_keys"], atol=1e-5)

    keys = T(d["keys"]).requires_grad_(True)
    feat = T(d["feat"]).requires_grad_(True)
    lc, idx = R.positions(keys, W, H, dim)
    o = R.slice_(lc, idx, R.splat(lc, idx, feat, pad, W, H, dim), pad, W, H, dim)
    np.testing.assert_allclose(o.detach().numpy(), d["chain_out"], atol=1e-6)
    (o * T(d["cot_o"])).sum().backward()
    np.testing.assert_allclose(feat.grad.numpy(), d["chain_g_feat"], atol=1e-5)
    np.testing.assert_allclose(keys.grad.numpy(), d["chain_g_keys"], atol=1e-5)


@pytest.mark.parametrize("case", sorted(load_golden("transforms")))
def test_transforms(case):
    d = load_golden("transforms")[case]
    dim = 2 if case.startswith("plane") else 3
    scales = T(d["scales"]) if "scales" in d else None
    y = R.rigid_transform(T(d["pcd"]), T(d["log_R"]), T(d["shift"]), scales, dim)
    np.testing.assert_allclose(y.numpy(), d["out"], atol=1e-6)


def _sd(d):
    return {k[3:]: T(v) for k, v in d.items() if k.startswith("sd/")}


BLOCK_CFG = {
    "mh2d": dict(in_feature_dim=4, tensor_size=16, tensor_dim=2, heads=4),
    "mh3d": dict(in_feature_dim=4, tensor_size=8, tensor_dim=3, heads=2),
    "mh2d_pad": dict(in_feature_dim=4, tensor_size=16, tensor_dim=2, heads=4),
    "pool": dict(in_feature_dim=4, tensor_size=8, tensor_dim=3, heads=2, pool=True),
    "adain": dict(in_feature_dim=4, tensor_size=16, tensor_dim=2, heads=4),
}
UNION_CFG = dict(features_dims=[4, 4], tensor_sizes=[16, 8], tensor_dims=[2, 3], heads=[4, 2])


@pytest.mark.parametrize("case", sorted(BLOCK_CFG))
def test_multihead_blocks(case):
    d = load_golden("blocks")[case]
    sd = _sd(d)
    x, pcd = T(d["x"]), T(d["pcd"])
    pad = T(d["pad"]) if "pad" in d else None
    style = T(d["style"]) if "style" in d else None
    modes = [m for m in ("eval", "train") if f"{m}_out" in d]
    for mode in modes:
        res, occ, mean, var = R.multihead(sd, x, pcd, train=(mode == "train"), pad=pad,
                                          style=style, **BLOCK_CFG[case])
        np.testing.assert_allclose(res.numpy(), d[f"{mode}_out"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(float(occ), d[f"{mode}_occ"][0], rtol=1e-6)
        if f"{mode}_mean" in d:
            np.testing.assert_allclose(float(mean), d[f"{mode}_mean"][0], atol=1e-6)
            np.testing.assert_allclose(float(var), d[f"{mode}_var"][0], rtol=1e-5)


@pytest.mark.parametrize("case", ["union", "union_proj", "union_adain"])
def test_union_blocks(case):
    d = load_golden("blocks")[case]
    sd = _sd(d)
    x, pcd = T(d["x"]), T(d["pcd"])
    style = T(d["style"]) if "style" in d else None
    modes = [m for m in ("eval", "train") if f"{m}_out" in d]
    for mode in modes:
        res, occs = R.multihead_union(sd, x, pcd, train=(mode == "train"), style=style, **UNION_CFG)
        np.testing.assert_allclose(res.numpy(), d[f"{mode}_out"], atol=3e-5, rtol=1e-5)
        np.testing.assert_allclose(np.array([float(o) for o in occs]), d[f"{mode}_occ"], rtol=1e-6)


def test_chamfer_against_reference_torch_restatement():
    """chamfer_extension/chamfer_pytorch.py:4-14 returns (P.min(1), P.min(2)) with
    P[b,i,j] = |a_i|^2+|b_j|^2-2a_i.b_j : ret0[j] = min_i (cloud-2 side), ret1[i] = min_j."""
    raw = np.load("tests/golden/chamfer.npz") if False else None
    import os
    from tests.conftest import GOLDEN
    raw = np.load(os.path.join(GOLDEN, "chamfer.npz"))
    a, b = T(raw["xyz1"]), T(raw["xyz2"])
    d1, d2, i1, i2 = R.chamfer_fwd(a, b)
    np.testing.assert_allclose(d2.numpy(), raw["ret0"], atol=2e-6)
    np.testing.assert_allclose(d1.numpy(), raw["ret1"], atol=2e-6)
    # argmins really attain the minima
    g = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
    assert torch.equal(g.gather(2, i1.long()[..., None])[..., 0], d1)
    assert torch.equal(g.gather(1, i2.long()[:, None])[:, 0], d2)
    # gradients of the reference's formulation under autograd: the cotangent of a min flows to its arg-min element, so
    # these pin the oracle's INDICES and its gradient formula (chamfer.cu:155-174) at once
    ga, gb = R.chamfer_bwd(a, b, T(raw["cot1"]), T(raw["cot0"]), i1, i2)
    np.testing.assert_allclose(ga.numpy(), raw["g_xyz1"], atol=2e-5)
    np.testing.assert_allclose(gb.numpy(), raw["g_xyz2"], atol=2e-5)

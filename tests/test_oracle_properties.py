"""Property tests of the CPU oracle (hypothesis): invariants of the op that hold for any input
 — they guard the restatement the GPU parity tests are measured against."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import ref_cpu as R

grid2 = st.tuples(st.integers(2, 20), st.integers(2, 20))
grid3 = st.tuples(st.integers(2, 9), st.integers(2, 9), st.integers(2, 9))


def _keys(seed, B, H, dim, N, spread):
    g = torch.Generator().manual_seed(seed)
    return torch.tanh(torch.randn(B, H * dim, N, generator=g) * spread)


@settings(max_examples=40, deadline=None)
@given(st.one_of(grid2, grid3), st.integers(0, 10 ** 6), st.floats(0.1, 4.0))
def test_weights_partition_unity_and_indices_in_range(W, seed, spread):
    dim = len(W)
    keys = _keys(seed, 2, 3, dim, 50, spread)
    keys[0, :, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.99999994])          # clamp edges
    lc, idx = R.positions(keys, W, 3, dim)
    assert lc.min() >= 0.0 and lc.max() <= 1.0
    np.testing.assert_allclose(lc.sum(2).numpy(), 1.0, atol=2e-6)
    assert idx.min() >= 0 and idx.max() < int(np.prod(W))
    # the 2^dim corners of a point are distinct cells
    s, _ = idx.sort(dim=2)
    assert (s[:, :, 1:] != s[:, :, :-1]).all()


@settings(max_examples=25, deadline=None)
@given(grid2, st.integers(0, 10 ** 6))
def test_splat_max_zero_floor_and_relu_idempotence(W, seed):
    g = torch.Generator().manual_seed(seed)
    keys = _keys(seed, 1, 2, 2, 64, 1.0)
    feat = torch.randn(1, 2 * 3, 64, generator=g)
    lc, idx = R.positions(keys, W, 2, 2)
    z = R.splat(lc, idx, feat, None, W, 2, 2, "max")
    assert z.min() >= 0.0
    assert torch.equal(z, R.splat(lc, idx, torch.relu(feat), None, W, 2, 2, "max"))
    # every grid value is one of the contributions (or the floor)
    contrib = (feat.reshape(1, 2, 3, 1, 64) * lc[:, :, None]).reshape(1, 2, 3, -1)
    zf = z.reshape(1, 2, 3, -1)
    for c in range(3):
        vals = set(contrib[0, 0, c].tolist()) | {0.0}
        assert all(v in vals for v in zf[0, 0, c].tolist())


@settings(max_examples=25, deadline=None)
@given(st.one_of(grid2, grid3), st.integers(0, 10 ** 6))
def test_sum_mode_mass_and_slice_of_constant(W, seed):
    dim = len(W)
    keys = _keys(seed, 1, 2, dim, 40, 1.5)
    lc, idx = R.positions(keys, W, 2, dim)
    ones = torch.ones(1, 2 * 2, 40)
    mass = R.splat(lc, idx, ones, None, W, 2, dim, "sum").reshape(1, 4, -1).sum(-1)
    np.testing.assert_allclose(mass.numpy(), 40.0, rtol=1e-5)             # weights sum to one per point
    const = torch.full((1, 4, *W), 2.5)
    out = R.slice_(lc, idx, const, None, W, 2, dim)
    np.testing.assert_allclose(out.numpy(), 2.5, rtol=1e-5)               # partition of unity


@settings(max_examples=15, deadline=None)
@given(grid2, st.integers(0, 10 ** 6))
def test_single_point_roundtrip_and_grad_balancing(W, seed):
    """Slice(Splat) of a single positive-feature point returns f * sum_v w_v^2; and the key gradient
    carries no (W-1)/2 factor (GradientBalancing, cloud_transform.py:12-26)."""
    g = torch.Generator().manual_seed(seed)
    keys = (torch.rand(1, 2, 1, generator=g) * 1.8 - 0.9).requires_grad_(True)
    f = torch.tensor([[[1.7]]])
    lc, idx = R.positions(keys, W, 1, 2)
    z = R.splat(lc, idx, f, None, W, 1, 2, "max")
    out = R.slice_(lc, idx, z, None, W, 1, 2)
    np.testing.assert_allclose(float(out), 1.7 * float((lc ** 2).sum()), rtol=1e-5)
    # d(lc corner 0)/d(key_x) = -w0y  (NOT -(W-1)/2 * w0y)
    (gk,) = torch.autograd.grad(lc[0, 0, 0, 0], keys)
    s = (keys.detach().clamp(-1 + 1e-7, 1 - 1e-7) + 1) * ((torch.tensor(W, dtype=torch.float32)[None, :, None] - 1) * 0.5)
    w0y = float((s.floor() + 1 - s)[0, 1, 0])
    np.testing.assert_allclose(float(gk[0, 0, 0]), -w0y, atol=1e-6)


@settings(max_examples=20, deadline=None)
@given(st.integers(1, 3), st.integers(1, 40), st.integers(1, 40), st.integers(0, 10 ** 6))
def test_chamfer_symmetry_and_selfdistance(B, n, m, seed):
    g = torch.Generator().manual_seed(seed)
    a, b = torch.rand(B, n, 3, generator=g), torch.rand(B, m, 3, generator=g)
    d1, d2, i1, i2 = R.chamfer_fwd(a, b)
    e2, e1, j2, j1 = R.chamfer_fwd(b, a)
    assert torch.equal(d1, e1) and torch.equal(d2, e2) and torch.equal(i1, j1) and torch.equal(i2, j2)
    s1, s2, k1, _ = R.chamfer_fwd(a, a.clone())
    assert float(s1.max()) == 0.0 and float(s2.max()) == 0.0


def test_fscore_oracle_known_answers():
    """oracle fscore (utils/f1_metric.py:9-30): hand-computed precision / recall, and the zero branches."""
    import numpy as np
    gt = np.array([[0., 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    pr = np.array([[0., 0, 0.005], [1, 0, 0.02], [5, 5, 5]])
    f, p, r = R.fscore(gt, pr, 0.01)
    assert (p, r) == (0.25, 1 / 3) and abs(f - 2 / 7) < 1e-12
    assert R.fscore(gt, gt, 0.01) == (1.0, 1.0, 1.0)
    assert R.fscore(gt, gt + 10, 0.01) == (0.0, 0.0, 0.0)
    assert R.fscore(gt, gt[:0], 0.01) == (0.0, 0.0, 0.0)

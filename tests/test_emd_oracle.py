"""CPU checks of the EMD oracle (oracle/emd_ref.c).  The reference has no known-answer
vectors for EMD; its only check (emd_linear/emd_module.py:79-93) recomputes the distance
from the returned assignment and looks at the number of distinct targets — applied here,
together with properties of the auction itself."""
import numpy as np
import pytest

from oracle import emd_ref


def _clouds(B, n, seed):
    rng = np.random.default_rng(seed)
    return rng.random((B, n, 3), dtype=np.float32), rng.random((B, n, 3), dtype=np.float32)


def test_reference_self_check_recipe():
    a, b = _clouds(2, 1024, 0)
    st, d, ass = emd_ref.forward(a, b, 0.005, 50)
    assert st == 1 and ass.min() >= 0 and ass.max() < 1024
    sel = np.take_along_axis(b, ass[..., None].astype(np.int64), axis=1)
    np.testing.assert_allclose(((a - sel) ** 2).sum(-1), d, atol=1e-7)          # "Verified EMD"
    assert all(len(np.unique(ass[i])) > 900 for i in range(2))                   # nearly a bijection


def test_converges_to_a_bijection_and_beats_random_matching():
    a, b = _clouds(1, 1024, 1)
    st, d, ass = emd_ref.forward(a, b, 0.002, 3000)
    assert len(np.unique(ass[0])) >= 1015          # near-bijection (the last iteration force-assigns)
    ident = np.sqrt(((a - b) ** 2).sum(-1)).mean()                               # identity matching
    assert np.sqrt(d).mean() < 0.25 * ident
    # a matching can never beat the nearest-neighbour lower bound
    full = ((a[0][:, None] - b[0][None]) ** 2).sum(-1)
    assert np.sqrt(d).mean() >= np.sqrt(full.min(1)).mean() - 1e-6


def test_identical_clouds_match_themselves():
    a, _ = _clouds(1, 1024, 2)
    st, d, ass = emd_ref.forward(a, a.copy(), 0.005, 50)
    assert np.array_equal(ass[0], np.arange(1024)) and float(d.max()) == 0.0


def test_preconditions():
    a, b = _clouds(1, 1000, 3)
    assert emd_ref.forward(a, b, 0.005, 5)[0] == -1                              # n % 1024 != 0


def test_backward_formula():
    a, b = _clouds(2, 1024, 4)
    _, d, ass = emd_ref.forward(a, b, 0.005, 10)
    g = np.random.default_rng(5).random((2, 1024), dtype=np.float32)
    ga = emd_ref.backward(a, b, g, ass)
    sel = np.take_along_axis(b, ass[..., None].astype(np.int64), axis=1)
    np.testing.assert_allclose(ga, 2 * g[..., None] * (a - sel), atol=1e-7)

"""The Chamfer oracle against THE REFERENCE'S OWN KERNELS: tests/golden/chamfer_reference.npz holds outputs of
chamfer_extension/chamfer.cu compiled for gfx950 (`make -C oracle ref_chamfer`) and run on an MI355X by
tests/golden/gen_chamfer_reference_golden.py through dist_chamfer.py's call sequence.  oracle/ref_cpu.py's chamfer_fwd / chamfer_bwd
(already pinned on the reference's pure-PyTorch chamfer_pytorch.py, tests/test_oracle_golden.py) must agree with the kernel too:
indices exactly on the lattice clouds (every tie exact: lowest target index, chamfer.cu:36,46,126) and wherever the nearest
target is unique to rounding on the uniform ones, distances to 1e-6 (on the lattice: exactly), gradients to 1e-5 (six float
atomics per pair in the reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chamfer_reference.npz")


@pytest.mark.parametrize("case", range(8))
def test_oracle_chamfer_equals_the_references_kernel(case):
    d = np.load(FIX)
    assert case < int(d["n_cases"])
    k = "c%02d_" % case
    a, b = torch.from_numpy(d[k + "xyz1"]), torch.from_numpy(d[k + "xyz2"])
    d1, d2, i1, i2 = R.chamfer_fwd(a, b)
    lattice = d[k + "kind"].item().decode() == "lattice"
    r1, r2 = torch.from_numpy(d[k + "dist1"]), torch.from_numpy(d[k + "dist2"])
    j1, j2 = torch.from_numpy(d[k + "idx1"]).long(), torch.from_numpy(d[k + "idx2"]).long()
    if lattice:
        assert torch.equal(i1.long(), j1) and torch.equal(i2.long(), j2)
        assert torch.equal(d1, r1) and torch.equal(d2, r2)
    else:
        assert float((d1 - r1).abs().max()) <= 1e-6 and float((d2 - r2).abs().max()) <= 1e-6
        full = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
        for mine, ref, table in ((i1.long(), j1, full), (i2.long(), j2, full.transpose(1, 2))):
            differ = mine != ref
            assert float(differ.float().mean()) < 1e-3
            # where they differ the two targets are equally near to rounding
            gap = (table.gather(2, mine[..., None]) - table.gather(2, ref[..., None]))[..., 0].abs()
            assert float(gap[differ].max() if differ.any() else 0.0) <= 1e-6
    ga, gb = R.chamfer_bwd(a, b, torch.from_numpy(d[k + "g1"]), torch.from_numpy(d[k + "g2"]), j1, j2)
    np.testing.assert_allclose(ga.numpy(), d[k + "g_xyz1"], atol=1e-5)
    np.testing.assert_allclose(gb.numpy(), d[k + "g_xyz2"], atol=1e-5)


def test_every_reference_target_of_the_oracle_makefile_has_a_rule():
    """`make -n ref` lists a recipe for each of the reference builds (the grid-subsampling rule was once lost in an edit and only
    a cold build of a clean checkout showed it)."""
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/emd_linear"):
        pytest.skip("the reference is not mounted here")
    r = subprocess.run(["make", "-C", os.path.join(here, "oracle"), "-n", "-B", "ref"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    for name in ("libgrid_subsampling_ref.so", "emd_reference.so", "emd_reference_strict.so", "chamfer_reference.so"):
        assert name in r.stdout, name

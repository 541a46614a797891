"""The blocks of the reference's import surface that none of its model_zoo files use — unet2d.unet_parts.{DoubleConv, Down,
Up, OutConv, GroupCat}, unet2d.unet_model.UNet, layers.v2v_groups.{EncoderDecorder, V2VModel}
(unet2d/unet_parts.py:49-150, unet2d/unet_model.py:8-41, layers/v2v_groups.py:73-171) — exist under the same import
paths, take the reference's state dicts strictly and compute the same function (CPU: the torch fall-back of the conv
modules; the reference side is imported from /root/reference in a child process, build container only)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_import_surface():
    sys.path.insert(0, ROOT)
    import unet2d.unet_parts as P
    import unet2d.unet_model as M
    import layers.v2v_groups as V
    for name in ("Res2DBlock", "Basic2DBlock", "DoubleConv", "Down", "Up", "OutConv", "GroupCat", "nn", "F", "torch"):
        assert hasattr(P, name), name
    assert hasattr(M, "UNet")
    for name in ("Basic3DBlock", "Res3DBlock", "Pool3DBlock", "Upsample3DBlock", "EncoderDecorder", "V2VModel"):
        assert hasattr(V, name), name
    gc = P.GroupCat(2)
    a, b = torch.arange(8.).reshape(1, 4, 1, 2), -torch.arange(4.).reshape(1, 2, 1, 2)
    out = gc(a, b)
    assert out.shape == (1, 6, 1, 2)
    assert torch.equal(out[0, :, 0, 0], torch.tensor([0., 2., -0., 4., 6., -2.]))


CHILD = r"""
import sys, torch
kind = sys.argv[1]
def build(paths):
    for m in [m for m in sys.modules if m.split(".")[0] in ("unet2d", "layers")]:
        del sys.modules[m]
    sys.path[:0] = paths
    try:
        if kind == "unet":
            from unet2d.unet_model import UNet
            torch.manual_seed(0)
            return UNet(2, 3, 16)
        from layers.v2v_groups import V2VModel
        torch.manual_seed(0)
        return V2VModel(2, 3, groups=2)
    finally:
        del sys.path[:len(paths)]
ref = build([%(ref)r])
assert ref.__class__.__module__.startswith(("unet2d", "layers")) and %(ref)r in sys.modules[ref.__class__.__module__].__file__
ours = build([%(root)r])
assert %(root)r in sys.modules[ours.__class__.__module__].__file__ or "cloud_transformers_amd" in ours.__class__.__module__
sd = ref.state_dict()
assert list(sd) == list(ours.state_dict()) or set(sd) == set(ours.state_dict())
# same seed -> same initial weights (module creation order matches)
same_init = all(torch.equal(v, ours.state_dict()[k]) for k, v in sd.items())
ours.load_state_dict(sd, strict=True)
torch.manual_seed(1)
x = torch.randn(2, 32, 32, 32) if kind == "unet" else torch.randn(1, 4, 16, 16, 16)
ref.eval(); ours.eval()
with torch.no_grad():
    a, b = ref(x), ours(x)
err = float((a - b).abs().max() / a.abs().max())
print("RESULT", same_init, err)
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (build container only)")
@pytest.mark.parametrize("kind", ["unet", "v2v"])
def test_matches_reference_module(kind):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "ref": REF}, kind], capture_output=True, text=True,
                       timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1] == "True", "initial weights differ from the reference's at the same seed"
    assert float(line[2]) <= 1e-5

"""Level-1 integration: the reference's own model_zoo files, exec'd the way utils/train_util.py:23-34
does, resolve `layers.*` / `unet2d.*` to THIS package and build models whose state-dict layout (every parameter /
buffer name and shape) is the one the reference's layers produce — released checkpoints load with strict=True.

The expected layouts are data recorded from the reference by tests/golden/gen_zoo_state_dicts.py.  The model files
themselves are read from /root/reference at test time, so this test only runs in the build container (it is skipped
where the reference is not mounted; it needs no GPU: construction and state dicts only)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %(root)r)            # this package's layers/ unet2d/ utils/ chamfer_extension/ emd_linear/ first
sys.path.append(%(ref)r)                # the reference last: only model_zoo/ (and its harness) come from it
import layers.multihead_ct as L
assert L.__file__.startswith(%(root)r), L.__file__
ns = {"__name__": "zoo_model"}
exec(compile(open(%(ref)r + "/" + sys.argv[1]).read(), sys.argv[1], "exec"), ns)
model = ns["Model"]()
sd = model.state_dict()
# a reference-layout state dict (same keys / shapes, fresh values) must load strictly
fake = {k: torch.zeros_like(v) for k, v in sd.items()}
model.load_state_dict(fake, strict=True)
print(json.dumps({k: list(v.shape) for k, v in sd.items()}))
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (build container only)")
@pytest.mark.parametrize("rel", ["model_zoo/s3dis/segmenter.py", "model_zoo/s3dis/segmenter_pad.py",
                                 "model_zoo/scanobject/classifier.py", "model_zoo/scanobject/classifier_scales.py",
                                 "model_zoo/completion/inpainter.py"])
def test_reference_zoo_file_builds_on_this_package(rel):
    with open(os.path.join(ROOT, "tests", "golden", "zoo_state_dicts.json")) as f:
        expected = json.load(f)[rel]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "ref": REF}, rel], capture_output=True, text=True,
                       timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(got) == set(expected), (sorted(set(expected) - set(got))[:5], sorted(set(got) - set(expected))[:5])
    wrong = {k: (got[k], expected[k]) for k in got if got[k] != expected[k]}
    assert not wrong, list(wrong.items())[:5]


REF_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r + "/tests/golden")
import gen_golden as G
G._install_shims()
sys.path.insert(0, G.REF)
ns = {"__name__": "zoo_model"}
exec(compile(open(G.REF + "/" + sys.argv[1]).read(), sys.argv[1], "exec"), ns)
torch.manual_seed(0)
torch.save(ns["Model"]().state_dict(), sys.argv[2])
"""

OURS_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r)
sys.path.append(%(ref)r)
ns = {"__name__": "zoo_model"}
exec(compile(open(%(ref)r + "/" + sys.argv[1]).read(), sys.argv[1], "exec"), ns)
torch.manual_seed(0)
sd = ns["Model"]().state_dict()
ref = torch.load(sys.argv[2])
bad = [k for k in sd if not torch.equal(sd[k], ref[k])]
print("DIFFERING", len(bad), bad[:5])
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (build container only)")
@pytest.mark.parametrize("rel", ["model_zoo/s3dis/segmenter.py", "model_zoo/completion/inpainter.py"])
def test_same_seed_same_weights(rel, tmp_path):
    """This package's modules draw their initial parameters in the reference's order: after torch.manual_seed(s) a zoo
    model built here has bit-identical weights to the one built on the reference's layers.  (That is what lets the
    whole-model GPU parity test, tests/test_zoo_gpu.py, rebuild the golden's weights from a seed instead of a 37 MB file.)"""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    sd_path = str(tmp_path / "ref_sd.pt")
    a = subprocess.run([sys.executable, "-c", REF_CHILD % {"root": ROOT}, rel, sd_path], capture_output=True, text=True,
                       timeout=300, env=env, cwd="/tmp")
    assert a.returncode == 0, a.stderr[-3000:]
    b = subprocess.run([sys.executable, "-c", OURS_CHILD % {"root": ROOT, "ref": REF}, rel, sd_path], capture_output=True,
                       text=True, timeout=300, env=env, cwd="/tmp")
    assert b.returncode == 0, b.stderr[-3000:]
    assert "DIFFERING 0 " in b.stdout, b.stdout[-500:]


REF_CKPT_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r + "/tests/golden")
import gen_golden as G
G._install_shims()
sys.path.insert(0, G.REF)
import types
sys.modules.setdefault("tensorboardX", types.SimpleNamespace(SummaryWriter=object))      # imported by utils/train_util.py, unused here
from utils.train_util import save_exp                       # the REFERENCE's checkpoint writer (utils/train_util.py:74-79)
ns = {"__name__": "zoo_model"}
exec(compile(open(G.REF + "/" + sys.argv[1]).read(), sys.argv[1], "exec"), ns)
torch.manual_seed(123)
model = ns["Model"]()
with torch.no_grad():
    for p in model.parameters():
        p.add_(torch.randn_like(p) * 0.01)                  # a "trained" checkpoint: nothing at its initial value
    for b in model.buffers():
        if b.is_floating_point():
            b.add_(torch.rand_like(b))
wrapped = torch.nn.DataParallel(model)                      # released weights carry the `module.` prefix (train_*.py wrap in DDP)
save_exp([wrapped], ["model"], sys.argv[2], 7)
"""

OURS_CKPT_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r)
sys.path.append(%(ref)r)
from utils.train_util import restore_exp_fix                # THIS package's harness behind the reference's import path
import utils.train_util as T
assert %(root)r in T.__file__, T.__file__
ns = {"__name__": "zoo_model"}
exec(compile(open(%(ref)r + "/" + sys.argv[1]).read(), sys.argv[1], "exec"), ns)
torch.manual_seed(0)
model = ns["Model"]()
path = sys.argv[2] + "/model_epoch_7.t7"
restore_exp_fix([model], [path], device=torch.device("cpu"), verbose=False)      # strict=True inside
ref = {k[7:]: v for k, v in torch.load(path).items()}
sd = model.state_dict()
bad = [k for k in sd if not torch.equal(sd[k], ref[k])]
print("LOADED", len(sd), "DIFFERING", len(bad), bad[:5])
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (build container only)")
@pytest.mark.parametrize("rel", ["model_zoo/s3dis/segmenter.py", "model_zoo/scanobject/classifier.py", "model_zoo/completion/inpainter.py"])
def test_reference_written_checkpoint_loads_strictly(rel, tmp_path):
    """A `.t7` written by the REFERENCE's own `save_exp` from a model on the REFERENCE's layers (DataParallel-wrapped, so
    with the `module.` prefix the released weights carry: configs/eval/*.yaml:2) loads with strict=True through this
    package's `utils.train_util.restore_exp_fix` into the same model_zoo file built on THIS package's layers, every tensor
    equal.  (The released checkpoints themselves cannot be fetched: no network.)"""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    a = subprocess.run([sys.executable, "-c", REF_CKPT_CHILD % {"root": ROOT}, rel, str(tmp_path)], capture_output=True, text=True,
                       timeout=300, env=env, cwd="/tmp")
    assert a.returncode == 0, a.stderr[-3000:]
    b = subprocess.run([sys.executable, "-c", OURS_CKPT_CHILD % {"root": ROOT, "ref": REF}, rel, str(tmp_path)], capture_output=True,
                       text=True, timeout=300, env=env, cwd="/tmp")
    assert b.returncode == 0, b.stderr[-3000:]
    assert "DIFFERING 0 " in b.stdout and "LOADED" in b.stdout, b.stdout[-500:]


def test_convert_pointwise_keeps_parameters_and_keys():
    """layers.pointwise.convert_pointwise: plain nn.Conv1d(k=1) layers of a model file become PointwiseConv1d in place — same
    parameter objects, same state-dict keys, other convolutions untouched; on CPU tensors the layer still runs torch's conv."""
    import torch
    from torch import nn
    from cloud_transformers_amd.layers.pointwise import PointwiseConv1d, convert_pointwise
    m = nn.Sequential(nn.Conv1d(6, 8, 1), nn.Conv1d(8, 8, 3, padding=1), nn.Conv1d(8, 4, 1, bias=False), nn.Conv1d(4, 4, 1, groups=2))
    params = [p for p in m.parameters()]
    keys = list(m.state_dict().keys())
    x = torch.randn(2, 6, 10)
    want = m(x)
    assert convert_pointwise(m) is m
    assert [type(l) for l in m] == [PointwiseConv1d, nn.Conv1d, PointwiseConv1d, nn.Conv1d]
    assert all(a is b for a, b in zip(params, m.parameters())) and list(m.state_dict().keys()) == keys
    assert torch.equal(m(x), want)

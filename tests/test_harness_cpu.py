"""The training harness on the CPU: config schema, model-file loading, optimizer / scheduler factories,
`.t7` checkpoint names and round trips (incl. the `module.` prefix of DistributedDataParallel checkpoints), and the
loop itself on a toy segmentation model."""
import os

import pytest
import torch
import yaml

from cloud_transformers_amd import harness as H

MODEL = '''
import torch
from torch import nn


class Model(nn.Module):
    def __init__(self, n_classes=8, dim=16):
        super().__init__()
        self.net = nn.Sequential(nn.Conv1d(3, dim, 1), nn.BatchNorm1d(dim), nn.ReLU(), nn.Conv1d(dim, n_classes, 1))
        self.scale = nn.Parameter(torch.zeros(1))

    def forward(self, cloud):                      # (B, 3, 1, N) -> ((B, K, 1, N), lattice statistics)
        return (self.net(cloud.squeeze(2)) * (1 + self.scale)).unsqueeze(2), []
'''

CONFIG = '''
experiment:
    root: '{root}/exp'
    writer_root: '{root}/runs'
data:
    batch_size: 4
    batch_size_val: 4
    num_workers: 0
    num_points: 64
model:
    generator: '{root}/toy_model.py'
    n_classes: 8
    dim: 16
train:
    num_epochs: 2
    show_each: 2
    save_each: 3
    scale_lr: !!float 1e-2
    optimizer:
        type: 'Adam'
        lr: !!float 1e-2
        betas: [!!float 0.9, !!float 0.999]
        weight_decay: !!float 0
    scheduler:
        type: 'StepLR'
        gamma: !!float 0.7
        step_size: 4
'''


@pytest.fixture
def cfg_path(tmp_path):
    (tmp_path / "toy_model.py").write_text(MODEL)
    p = tmp_path / "toy.yaml"
    p.write_text(CONFIG.format(root=str(tmp_path)))
    return p


def test_factories_leave_the_config_reusable():
    net = torch.nn.Linear(3, 3)
    cfg = {"type": "Adam", "lr": 1e-3, "betas": [0.9, 0.999], "weight_decay": 0.0}
    opt = H.make_optimizer(net.parameters(), cfg)
    assert isinstance(opt, torch.optim.Adam) and cfg["type"] == "Adam"
    sch = H.make_scheduler(opt, {"type": "StepLR", "gamma": 0.7, "step_size": 25000})
    assert isinstance(sch, torch.optim.lr_scheduler.StepLR) and sch.gamma == 0.7


def test_loop_runs_from_a_yaml_config_and_writes_t7_checkpoints(cfg_path):
    cfg = H.load_config(cfg_path)
    assert cfg["train"]["optimizer"]["lr"] == 1e-2 and cfg["model"]["generator"].endswith("toy_model.py")
    tr = H.Trainer(cfg, "segmentation", n_classes=8, device=torch.device("cpu"), dataset_length=16)
    assert len(tr.optimizer.param_groups) == 2 and tr.optimizer.param_groups[1]["lr"] == 1e-2     # `scale` has its own rate
    hist = tr.fit()
    assert len(hist) == 8 and all(map(lambda v: v == v and v < 1e3, hist))
    assert hist[-1] < hist[0]                                     # labels follow the geometry: the loss moves
    assert tr.optimizer.param_groups[0]["lr"] == pytest.approx(1e-2 * 0.7 ** 2)
    files = sorted(os.listdir(tr.exp_dir))
    assert "toy_model.py" in files and "toy.yaml" in files        # model file and config are kept with the experiment
    assert "generator_iter_3.t7" in files and "g_opt_iter_6.t7" in files
    # checkpoints are plain state dicts that load strictly into a fresh model
    fresh = H.get_model(cfg["model"]["generator"], {"n_classes": 8, "dim": 16})
    H.restore_exp([fresh], [os.path.join(tr.exp_dir, "generator_iter_6.t7")], torch.device("cpu"), verbose=False)
    state = torch.load(os.path.join(tr.exp_dir, "generator_iter_6.t7"))
    assert list(state) == list(fresh.state_dict())


def test_restore_section_and_ddp_prefixed_checkpoints(cfg_path, tmp_path):
    cfg = H.load_config(cfg_path)
    net = H.get_model(cfg["model"]["generator"], {"n_classes": 8, "dim": 16})
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.25)
    ckpt = tmp_path / "generator_iter_7.t7"
    torch.save({"module." + k: v for k, v in net.state_dict().items()}, str(ckpt))      # as saved from a DDP wrapper
    opt = H.make_optimizer(net.parameters(), cfg["train"]["optimizer"])
    H.save_exp([opt], ["g_opt"], tmp_path, 7, epoch_name="iter")
    cfg["restore"] = {"generator": str(ckpt), "optimizer": str(tmp_path / "g_opt_iter_7.t7"), "new_lr": 5e-4}
    del cfg["train"]["scale_lr"]
    tr = H.Trainer(cfg, "segmentation", n_classes=8, device=torch.device("cpu"), dataset_length=8, make_dirs=False)
    for a, b in zip(tr.model.state_dict().values(), net.state_dict().values()):
        assert torch.equal(a, b)
    assert all(g["lr"] == 5e-4 for g in tr.optimizer.param_groups)
    with pytest.raises(RuntimeError):                              # strict: a missing key is an error
        bad = {"module." + k: v for k, v in list(net.state_dict().items())[1:]}
        torch.save(bad, str(ckpt))
        H.restore_exp_fix([net], [str(ckpt)], device=torch.device("cpu"), verbose=False)


def test_reference_import_path():
    from utils import train_util
    assert train_util.restore_exp_fix is H.restore_exp_fix and train_util.make_scheduler is H.make_scheduler


def test_synthetic_batches_have_the_loaders_shapes():
    seg = H.SyntheticClouds("segmentation", 128, 13, length=4, channels=6)
    pts, lab = seg[1]
    assert pts.shape == (128, 6) and pts.dtype == torch.float32 and lab.shape == (128,) and lab.dtype == torch.int64
    assert torch.equal(seg[1][0], pts) and int(lab.max()) < 13
    cls = H.SyntheticClouds("classification", 64, 15, length=4)
    pts, lab, mask = cls[0]
    assert pts.shape == (64, 3) and lab.shape == () and mask.shape == (64,) and mask.dtype == torch.float32
    comp = H.SyntheticClouds("completion", 1024, 256, length=2)
    noise, part, gt = comp[1]
    assert noise.shape == (4, 1024) and part.shape == (256, 3) and gt.shape == (1024, 3)
    assert torch.allclose(noise[:3].norm(dim=0), torch.ones(1024), atol=1e-5) and set(noise[3].unique().tolist()) <= {0.0, 1.0}


def test_data_kind_selects_the_dataset_readers(tmp_path):
    """`data.kind` of the config (or Trainer(dataset=...)) puts data/datasets.py's readers behind the loop: a ScanObjectNN
    file written here in the loader's npz form drives a classification step on the CPU."""
    import numpy as np
    n, pts = 12, 64
    rng = np.random.RandomState(0)
    np.savez(tmp_path / "scan.npz", data=rng.randn(n, pts, 3).astype(np.float32), label=rng.randint(0, 5, n).astype(np.int64),
             mask=(rng.rand(n, pts) > 0.5).astype(np.int64) - 1)
    (tmp_path / "cls_model.py").write_text('''
import torch
from torch import nn


class Model(nn.Module):
    def __init__(self, n_classes=5):
        super().__init__()
        self.f = nn.Conv1d(3, 8, 1)
        self.c = nn.Linear(8, n_classes)
        self.m = nn.Conv1d(8, 1, 1)

    def forward(self, cloud):
        h = torch.relu(self.f(cloud.squeeze(2)))
        return self.c(h.mean(2)), self.m(h)
''')
    cfg = yaml.safe_load(CONFIG.format(root=str(tmp_path)))
    cfg["model"] = {"generator": str(tmp_path / "cls_model.py"), "n_classes": 5}
    cfg["data"].update({"kind": "scanobjectnn", "path": str(tmp_path / "scan.h5"), "num_points": 32, "batch_size": 4})
    cfg["train"].pop("scale_lr")
    cfg["train"]["num_epochs"] = 1
    tr = H.Trainer(cfg, "classification", n_classes=5, device=torch.device("cpu"), make_dirs=False)
    from cloud_transformers_amd.data.datasets import ScanObjectNN
    assert isinstance(tr.loader.dataset, ScanObjectNN)
    hist = tr.fit(max_iters=3, log_each=1)
    assert len(hist) == 3 and all(np.isfinite(hist))
    with pytest.raises(ValueError):
        H.make_dataset({"data": {"kind": "kitti", "num_points": 8}}, "segmentation", 3)

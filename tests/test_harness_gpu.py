"""The training harness on the GPU: the S3DIS-shaped segmenter driven from a YAML config for a few steps on the HIP
kernels, `.t7` checkpoint written and loaded back strictly."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

MODEL = "from tests.test_zoo_gpu import Segmenter as Model\n"
CONFIG = '''
experiment:
    root: '{root}/exp'
    writer_root: '{root}/runs'
data:
    batch_size: 2
    num_workers: 0
    num_points: 512
model:
    generator: '{root}/segmenter.py'
    n_classes: 13
train:
    num_epochs: 1
    save_each: 3
    optimizer:
        type: 'Adam'
        lr: !!float 1e-3
        betas: [!!float 0.9, !!float 0.999]
        weight_decay: !!float 0.0
    scheduler:
        type: 'StepLR'
        gamma: !!float 0.7
        step_size: 25000
'''


def test_segmenter_trains_from_a_yaml_config_and_round_trips_its_checkpoint(tmp_path):
    from cloud_transformers_amd import harness as H
    (tmp_path / "segmenter.py").write_text(MODEL)
    cfg_path = tmp_path / "s3dis.yaml"
    cfg_path.write_text(CONFIG.format(root=str(tmp_path)))
    torch.manual_seed(0)
    tr = H.Trainer(H.load_config(cfg_path), "segmentation", n_classes=13, device=torch.device("cuda", 0), dataset_length=8,
                   channels=6)
    hist = tr.fit(max_iters=4)
    assert len(hist) == 4 and all(v == v for v in hist)
    assert min(hist[1:]) < hist[0]                      # Adam on geometric labels: the loss moves within a few steps
    ckpt = os.path.join(tr.exp_dir, "generator_iter_3.t7")
    assert os.path.exists(ckpt) and os.path.exists(os.path.join(tr.exp_dir, "g_opt_iter_3.t7"))
    fresh = H.get_model(str(tmp_path / "segmenter.py"), {"n_classes": 13}).cuda()
    H.restore_exp_fix([fresh], [ckpt], device=torch.device("cuda", 0), verbose=False)
    state = torch.load(ckpt)
    assert list(state) == list(fresh.state_dict())


COMPLETION = '''
experiment:
    root: '{root}/exp'
    writer_root: '{root}/runs'
data:
    batch_size: 2
    num_workers: 0
    num_points: 1024
model:
    generator: '{root}/inpainter.py'
train:
    num_epochs: 1
    chamfer_weight: !!float 1.0
    scale_lr: !!float 1e-2
    optimizer:
        type: 'Adam'
        lr: !!float 1e-4
        betas: [!!float 0.9, !!float 0.999]
        weight_decay: !!float 0.0
'''


def test_completion_task_steps_the_inpainter_with_emd_and_chamfer(tmp_path):
    """train_inpainter.py's step (186-195): EMD (auction, 50 iterations) + Chamfer loss through the AdaIN decoder; the
    residual scales of the AdaIN blocks get their own learning rate (`train.scale_lr`)."""
    from cloud_transformers_amd import harness as H
    (tmp_path / "inpainter.py").write_text("from tests.test_zoo_gpu import Inpainter as Model\n")
    cfg_path = tmp_path / "inpainting.yaml"
    cfg_path.write_text(COMPLETION.format(root=str(tmp_path)))
    torch.manual_seed(0)
    tr = H.Trainer(H.load_config(cfg_path), "completion", n_classes=256, device=torch.device("cuda", 0), dataset_length=4)
    assert len(tr.optimizer.param_groups) == 2 and len(tr.optimizer.param_groups[1]["params"]) > 0
    hist = tr.fit(max_iters=2)
    assert len(hist) == 2 and all(v == v and 0.0 < v < 10.0 for v in hist)


def test_graph_captured_step_follows_the_eager_trajectory(tmp_path):
    """`fit(hip_graph=True)`: forward + loss + backward replayed as one HIP graph on static input buffers, losses read back
    once per logging interval — the loss history must follow the eager run's (same seed, same batches; the captured step's
    warm-up passes must not leak into the running statistics or the weights)."""
    from cloud_transformers_amd import harness as H
    (tmp_path / "segmenter.py").write_text(MODEL)
    cfg_path = tmp_path / "s3dis.yaml"
    cfg_path.write_text(CONFIG.format(root=str(tmp_path)).replace("save_each: 3", "save_each: 100000"))
    hists = []
    for graph in (False, True):
        torch.manual_seed(0)
        tr = H.Trainer(H.load_config(cfg_path), "segmentation", n_classes=13, device=torch.device("cuda", 0), dataset_length=16,
                       channels=6, make_dirs=False)
        hists.append(tr.fit(max_iters=6, hip_graph=graph, log_each=4))
        assert len(hists[-1]) == 6
    eager, graphed = hists
    assert abs(eager[0] - graphed[0]) <= 1e-4 * abs(eager[0])                  # first step: identical weights and inputs
    assert all(abs(a - b) <= 2e-2 * abs(a) for a, b in zip(eager, graphed)), (eager, graphed)      # then the same trajectory


def test_graphed_fit_survives_a_ragged_last_batch(tmp_path):
    """`data.drop_last: false` with a dataset that is not a multiple of the batch: the last batch of every epoch has another
    shape.  It gets a graph of its own; every graph owns the gradient tensors it writes and the step points the parameters'
    `.grad` at the replayed graph's — the run must keep training after the odd batch (an eager fallback that dropped the
    captured gradient tensors would leave every later step a no-op) and follow the eager trajectory."""
    from cloud_transformers_amd import harness as H
    (tmp_path / "segmenter.py").write_text(MODEL)
    cfg_path = tmp_path / "s3dis.yaml"
    text = CONFIG.format(root=str(tmp_path)).replace("save_each: 3", "save_each: 100000").replace("num_epochs: 1", "num_epochs: 3")
    cfg_path.write_text(text.replace("    num_points: 512", "    num_points: 512\n    drop_last: false"))
    hists, finals = [], []
    for graph in (False, True):
        torch.manual_seed(0)
        tr = H.Trainer(H.load_config(cfg_path), "segmentation", n_classes=13, device=torch.device("cuda", 0), dataset_length=5,
                       channels=6, make_dirs=False)                     # batches of 2, 2, 1 per epoch
        hists.append(tr.fit(hip_graph=graph, log_each=4))
        assert len(hists[-1]) == 9
        if graph:
            assert len(tr._graphs) == 2
        finals.append(torch.cat([p.detach().flatten() for p in tr.model.parameters()]))
    eager, graphed = hists
    assert all(abs(a - b) <= 2e-2 * abs(a) for a, b in zip(eager, graphed)), (eager, graphed)
    # the weights moved in every epoch, also after the odd batch: both runs end close to each other and far from the start
    torch.manual_seed(0)
    start = torch.cat([p.detach().flatten() for p in H.Trainer(H.load_config(cfg_path), "segmentation", n_classes=13,
                                                               device=torch.device("cuda", 0), dataset_length=5, channels=6,
                                                               make_dirs=False).model.parameters()])
    moved = float((finals[1] - start).norm())
    assert moved > 0 and float((finals[1] - finals[0]).norm()) <= 0.2 * moved, (moved, float((finals[1] - finals[0]).norm()))

"""The reference's data-parallel recipe (train_segmentation.py:58-61,128-130:
init_process_group('nccl') + DistributedDataParallel(SyncBatchNorm.convert_sync_batchnorm(model)))
on the HIP modules: RCCL process group of one rank on the test box — checks that DDP's
autograd hooks fire through the ctypes-backed autograd Functions and that gradients equal the
plain run (multi-rank logic is covered on gloo in tests/test_parallel_gloo.py)."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_ddp_syncbn_single_rank_matches_plain():
    from cloud_transformers_amd.layers.multihead_ct import MultiHeadUnion
    from cloud_transformers_amd import parallel as P
    torch.manual_seed(0)
    model = MultiHeadUnion(32, [4, 4], [16, 8], [2, 3], [4, 2]).cuda()
    x = torch.randn(2, 32, 256, device="cuda")
    pcd = torch.rand(2, 3, 256, device="cuda") * 2 - 1
    out, _ = model(x, pcd)
    out.square().mean().backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()

    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        ddp = P.data_parallel(model, 0)
        assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in ddp.modules())
        out, stats = ddp(x, pcd)
        loss = out.square().mean()
        loss.backward()
        for n, p in ddp.module.named_parameters():
            assert p.grad is not None, n
            assert torch.allclose(p.grad, ref[n], atol=2e-5, rtol=1e-4), n
        red = P.reduce_loss_dict(dist, {"loss": loss.detach()})
        assert torch.allclose(red["loss"], loss.detach())
        assert P.max_over_ranks(dist, 1.5) == 1.5
        P.barrier(dist)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (the build's gpurun boxes have one)")
def test_two_rank_ddp_step_on_rccl():
    """So that the first 8-GPU run is not also the first RCCL run with more than one rank: `bench.py --gpus 2 --mode
    ddp-step` through launch.spawn_ranks (one process per GPU, RCCL over xGMI) — both ranks must be seen, the fused
    SyncBatchNorm groups must issue their 76 statistics collectives per step (6 per MultiHeadUnion block x 12 + 2 x 2 for
    the stock norms of stem and head), and the ranks must end the steps with identical parameters and running statistics.
    Reference recipe: train_segmentation.py:58-61,128-130."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "ddp-step", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["world_size_seen"] == 2
    assert cfg["norm_statistics_collectives_per_step"] == 76
    assert cfg["params_equal_across_ranks"], cfg["param_checksum_spread"]


def test_one_rank_ddp_step_graphed_with_the_statistics_exchange_forced_on():
    """`bench.py --mode ddp-step` at one rank, in a child process (its own RCCL group): forward + loss + backward of the
    S3DIS-shaped segmenter under DistributedDataParallel + SyncBatchNorm captured as ONE HIP graph, with the norms'
    statistics exchange forced on (CLOUDCT_SYNCBN_FORCE: a one-rank group would otherwise skip it) — the blocks' 72
    collectives per step and the bucketed gradient all-reduce a multi-rank step enqueues, captured and replayed; the eager
    step of the same process (--no-graph) must produce the same loss after the same number of steps."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for extra in ([], ["--no-graph"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "ddp-step", "--steps", "4", "--warmup", "2",
                            "--batch", "2", "--points", "1024"] + extra, capture_output=True, text=True, timeout=900)
        why = [l for l in r.stderr.splitlines() if not l.startswith("frame #") and l.strip()]
        assert r.returncode == 0, "\n".join(why[-25:])[-4000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    g, e = lines
    assert g["config"]["step"].startswith("HIP graph"), g["config"]["step"]
    assert e["config"]["step"].startswith("eager"), e["config"]["step"]
    for line in lines:
        # 6 per MultiHeadUnion block x 12; the 4 of the stem's / head's stock nn.SyncBatchNorm layers (76 in all at two ranks and
        # more, test_two_rank_ddp_step_on_rccl) are skipped by torch itself in a one-rank group
        assert line["config"]["norm_statistics_collectives_per_step"] == 72, line["config"]
        assert "forced" in line["config"]["norm_statistics_exchange"]
    assert abs(g["config"]["loss"] - e["config"]["loss"]) <= 2e-2 * abs(e["config"]["loss"]), (g["config"]["loss"], e["config"]["loss"])


def test_harness_fit_graphs_the_ddp_step(tmp_path):
    """harness.Trainer under DistributedDataParallel + SyncBatchNorm (one RCCL rank, the norms' statistics exchange forced
    on): fit(hip_graph=True) captures forward + loss + backward with the collectives and follows the eager trajectory."""
    from cloud_transformers_amd import harness as H
    from cloud_transformers_amd import ops
    from tests.test_harness_gpu import CONFIG, MODEL
    (tmp_path / "segmenter.py").write_text(MODEL)
    cfg_path = tmp_path / "s3dis.yaml"
    cfg_path.write_text(CONFIG.format(root=str(tmp_path)).replace("save_each: 3", "save_each: 100000"))
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    forced = ops.SYNC_STATS_FORCE
    ops.SYNC_STATS_FORCE = True
    try:
        hists = []
        for graph in (False, True):
            torch.manual_seed(0)
            tr = H.Trainer(H.load_config(cfg_path), "segmentation", n_classes=13, device=torch.device("cuda", 0), dist=dist,
                           dataset_length=16, channels=6, make_dirs=False)
            assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
            c0 = ops.sync_stats_collectives()
            hists.append(tr.fit(max_iters=5, hip_graph=graph, log_each=5))
            assert ops.sync_stats_collectives() > c0                   # the exchange path ran (captured once when graphed)
            assert len(hists[-1]) == 5
        eager, graphed = hists
        assert abs(eager[0] - graphed[0]) <= 1e-4 * abs(eager[0])
        assert all(abs(a - b) <= 2e-2 * abs(a) for a, b in zip(eager, graphed)), (eager, graphed)
    finally:
        ops.SYNC_STATS_FORCE = forced
        dist.destroy_process_group()

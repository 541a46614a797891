"""world_size-2 gloo tests of the N>1 host logic (no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cloud_transformers_amd import parallel as P
    res = {}
    P.barrier(dist)
    res["max"] = P.max_over_ranks(dist, 1.0 + rank)
    res["shard"] = P.shard_range(9, rank, world)
    ld = P.reduce_loss_dict(dist, {"b": torch.tensor(2.0 * (rank + 1)), "a": torch.tensor(1.0 * (rank + 1))})
    res["loss"] = {k: float(v) for k, v in ld.items()}
    g = P.all_gather_tensor(dist, torch.full((3,), float(rank)))
    res["gather"] = [t.tolist() for t in g]
    res["objects"] = P.all_gather(dist, {"rank": rank, "ids": list(range(rank + 1))})   # ragged payloads
    # the benchmark's aggregate: every rank processes its own clouds, value = units / max time
    per_rank_units = 8 * 4096
    t = P.max_over_ranks(dist, 0.5 if rank == 0 else 1.0)
    res["value"] = world * per_rank_units / t
    # data_parallel: same weights everywhere, each rank its own samples -> every rank ends with the mean gradient
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.ReLU(), torch.nn.Linear(3, 1))
    ddp = P.data_parallel(net, None, sync_bn=False)
    x = torch.full((2, 4), float(rank + 1))
    ddp(x).sum().backward()
    res["ddp_grad"] = net[0].weight.grad.flatten().tolist()
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.ReLU(), torch.nn.Linear(3, 1))
    (sum(ref(torch.full((2, 4), float(r + 1))).sum() for r in range(world)) / world).backward()
    res["ref_grad"] = ref[0].weight.grad.flatten().tolist()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_helpers():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert out[0]["max"] == out[1]["max"] == 2.0
    assert out[0]["shard"] == (0, 5) and out[1]["shard"] == (5, 9)
    assert out[0]["loss"] == {"a": 1.5, "b": 3.0}            # averaged on rank 0
    assert out[0]["gather"] == out[1]["gather"] == [[0.0] * 3, [1.0] * 3]
    assert out[0]["objects"] == out[1]["objects"] == [{"rank": 0, "ids": [0]}, {"rank": 1, "ids": [0, 1]}]
    assert out[0]["value"] == out[1]["value"] == 2 * 8 * 4096 / 1.0
    for r in range(world):
        assert out[r]["ddp_grad"] == pytest.approx(out[r]["ref_grad"], rel=1e-6, abs=1e-7)


def test_single_process_fallbacks():
    from cloud_transformers_amd import parallel as P
    assert P.max_over_ranks(None, 3.5) == 3.5
    assert P.world_size(None) == 1
    P.barrier(None)
    assert P.shard_range(10, 0, 1) == (0, 10)
    owned = [P.shard_range(10, r, 4) for r in range(4)]
    assert owned == [(0, 3), (3, 6), (6, 8), (8, 10)]
    d = {"x": torch.tensor(1.0)}
    assert P.reduce_loss_dict(None, d) is d


def test_reference_named_utils_package(tmp_path):
    """`utils.*` import paths of the reference resolve to this build (no process group needed)."""
    import utils.f1_metric as F
    import utils.grdnet_utils as G
    import utils.pcd_utils as U
    import utils.train_util_distributed as T
    assert callable(F.get_f1_scores) and callable(F.get_f1_scores_merge) and callable(F.calculate_fscore)
    assert G.Metrics.names() == ["F-Score", "ChamferDistance"] and callable(U.sphere_noise)
    assert T.all_gather("x") == ["x"]
    losses = {"a": torch.tensor(1.0)}
    assert T.reduce_loss_dict(losses) is losses
    # DDP-unwrapping save + restore round trip of a state dict
    net, net2 = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    T.save_exp_parallel([net], ["model"], tmp_path, 7)
    T.restore_exp([net2], [str(tmp_path / "model_epoch_7.t7")], device=torch.device("cpu"), verbose=False)
    assert torch.equal(net.weight, net2.weight) and torch.equal(net.bias, net2.bias)

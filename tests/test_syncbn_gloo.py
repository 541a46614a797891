"""Host logic of the statistics exchange of the fused norm groups (ops._bn_group_fwd / _bn_group_bwd) at world size 2
on gloo, without a GPU: the HIP kernels are replaced by oracle/bn_ref.FakeLib (float64 stand-ins on the same pointers),
everything else — buffer layout, channel offsets, ONE all_gather forward and ONE all_reduce backward per group, local
parameter gradients — is the product's code.  Reference: plain float64 batch norm over the whole batch, which is what
nn.SyncBatchNorm computes (train_segmentation.py:128 converts every norm of the reference's models)."""
import os
import socket

import pytest
import torch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ref_bn(x, w, b, eps, relu):
    mean = x.mean(dim=(0, 2), keepdim=True)
    var = x.var(dim=(0, 2), unbiased=False, keepdim=True)
    y = (x - mean) / torch.sqrt(var + eps) * w[None, :, None] + b[None, :, None]
    return torch.relu(y) if relu else y


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cloud_transformers_amd import _lib, ops
        from oracle.bn_ref import FakeLib
        fake = FakeLib()
        _lib.load = lambda: fake                     # the HIP kernels' stand-ins (CPU, float64 inside)
        _lib.check = lambda status, what: None if status == 0 else (_ for _ in ()).throw(RuntimeError(what))
        ops._dev = lambda *t: None                   # (the product refuses CPU tensors; this test is about the host logic)
        ops._stream = lambda: None
        g = torch.Generator().manual_seed(11)
        # ragged shards: rank 0 holds 1 cloud, rank 1 holds 3 (counts differ: the merge must weight by count)
        shards = [slice(0, 1), slice(1, 4)]
        Bg, N = 4, 64
        sl = shards[rank]
        Ck, Cv = 6, 10
        xg = torch.randn(Bg, Ck + Cv, N, generator=g) * 2 + 0.5
        cot = torch.randn(Bg, Ck + Cv, N, generator=g)

        def sbn(C, seed):
            torch.manual_seed(seed)
            m = torch.nn.SyncBatchNorm(C)
            with torch.no_grad():
                m.weight.copy_(torch.rand(C) + 0.5)
                m.bias.copy_(torch.randn(C) * 0.1)
            return m.train()

        bk, bv, ja, jb = sbn(Ck, 1), sbn(Cv, 2), sbn(Ck, 3), sbn(Cv, 4)
        assert ops._sync_group(bk) is not None
        before = ops.sync_stats_collectives()
        x = xg[sl].clone().requires_grad_(True)
        a, b = ops.split_bn(x, bk, bv)
        mid = ops.sync_stats_collectives()
        y = ops.join_bn_relu([a, b], [ja, jb])
        (y * cot[sl]).sum().backward()
        ncoll = ops.sync_stats_collectives() - before
        # reference on the whole batch in float64
        xr = xg.double().clone().requires_grad_(True)
        P = {m: (m.weight.detach().double().requires_grad_(True), m.bias.detach().double().requires_grad_(True)) for m in (bk, bv, ja, jb)}
        ar = _ref_bn(xr[:, :Ck], *P[bk], bk.eps, False)
        br = _ref_bn(xr[:, Ck:], *P[bv], bv.eps, False)
        yr = torch.cat([_ref_bn(ar, *P[ja], ja.eps, True), _ref_bn(br, *P[jb], jb.eps, True)], dim=1)
        (yr * cot.double()).sum().backward()
        res = {"y": float((y.detach().double() - yr.detach()[sl]).abs().max()),
               "gx": float((x.grad.double() - xr.grad[sl]).abs().max()),
               "collectives": ncoll, "after_split": mid - before}
        # parameter gradients are this rank's share; summed over the ranks they are the whole batch's
        for name, m in (("bk", bk), ("ja", ja)):
            gw = m.weight.grad.detach().clone()
            dist.all_reduce(gw)
            res["gw_" + name] = float((gw.double() - P[m][0].grad).abs().max())
        M = Bg * N
        res["running_var"] = float((bk.running_var.double() - (0.9 + 0.1 * xg[:, :Ck].double().var(dim=(0, 2), unbiased=False) * M / (M - 1))).abs().max())
        res["nbt"] = int(bk.num_batches_tracked)
        q.put((rank, res))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_norm_group_statistics_exchange_world_size_2():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank in range(world):
        r = got[rank]
        assert r["y"] <= 1e-5 and r["gx"] <= 1e-5, r
        assert r["gw_bk"] <= 1e-4 and r["gw_ja"] <= 1e-4, r
        assert r["running_var"] <= 1e-5 and r["nbt"] == 1, r
        assert r["after_split"] == 1                # two norms, one all_gather
        assert r["collectives"] == 4                # (split_bn + join_bn_relu) x (forward gather + backward reduce)

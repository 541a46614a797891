"""Fused norms under SyncBatchNorm: two ranks (gloo rendezvous, both on cuda:0 — the statistics exchange needs a
process group, not two GPUs) run the four fused norm groups of a union block on their shards; outputs, running
statistics and every gradient must equal plain float64 batch norm over the WHOLE batch, and each group must cost one
collective per direction."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ref_bn(x, w, b, eps, relu, residual=None):
    """float64 training-mode batch norm over the whole batch (+ ReLU, + skip)"""
    mean = x.mean(dim=(0, 2), keepdim=True)
    var = x.var(dim=(0, 2), unbiased=False, keepdim=True)
    y = (x - mean) / torch.sqrt(var + eps) * w[None, :, None] + b[None, :, None]
    if relu:
        y = torch.relu(y)
    return y if residual is None else y + residual


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cloud_transformers_amd import ops
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        g = torch.Generator().manual_seed(5)
        Bg, Cin, N = 4, 24, 256                       # global batch 4: two clouds per rank
        Bl = Bg // world
        sl = slice(rank * Bl, (rank + 1) * Bl)
        res = {}
        c0 = ops.sync_stats_collectives()

        def sync_bn(C, seed):
            torch.manual_seed(seed)
            bn = torch.nn.BatchNorm1d(C)
            with torch.no_grad():
                bn.weight.copy_(torch.rand(C) + 0.5)
                bn.bias.copy_(torch.randn(C) * 0.1)
            return torch.nn.SyncBatchNorm.convert_sync_batchnorm(bn).to(dev).train()

        # ---- 1. bn_relu with the skip connection
        C = 16
        xg = torch.randn(Bg, C, N, generator=g, dtype=torch.float64)
        rg = torch.randn(Bg, C, N, generator=g, dtype=torch.float64)
        cg = torch.randn(Bg, C, N, generator=g, dtype=torch.float64)
        bn = sync_bn(C, 1)
        assert ops.bn_relu_eligible(bn, xg[sl].float().to(dev))
        x = xg[sl].float().to(dev).requires_grad_(True)
        r = rg[sl].float().to(dev).requires_grad_(True)
        y = ops.bn_relu(x, bn, relu=True, residual=r)
        (y * cg[sl].float().to(dev)).sum().backward()
        xr = xg.clone().requires_grad_(True)
        w64, b64 = bn.weight.detach().double().cpu().requires_grad_(True), bn.bias.detach().double().cpu().requires_grad_(True)
        yr = _ref_bn(xr, w64, b64, bn.eps, True, rg)
        (yr * cg).sum().backward()
        res["bn_relu_y"] = float((y.detach().cpu().double() - yr.detach()[sl]).abs().max())
        res["bn_relu_gx"] = float((x.grad.cpu().double() - xr.grad[sl]).abs().max())
        # parameter gradients: this rank's share (DDP sums / averages them); the shares of all ranks add up to the whole
        gw = bn.weight.grad.detach().clone()
        gb = bn.bias.grad.detach().clone()
        dist.all_reduce(gw)
        dist.all_reduce(gb)
        res["bn_relu_gw"] = float((gw.cpu().double() - w64.grad).abs().max() / w64.grad.abs().max())
        res["bn_relu_gb"] = float((gb.cpu().double() - b64.grad).abs().max() / b64.grad.abs().max())
        M = Bg * N
        rm = 0.1 * xg.mean(dim=(0, 2))
        rv = 0.9 + 0.1 * xg.var(dim=(0, 2), unbiased=False) * M / (M - 1)
        res["running_mean"] = float((bn.running_mean.cpu().double() - rm).abs().max())
        res["running_var"] = float((bn.running_var.cpu().double() - rv).abs().max())
        res["nbt"] = int(bn.num_batches_tracked)
        res["collectives_bn_relu"] = ops.sync_stats_collectives() - c0

        # ---- 2. split_bn (two norms on the halves of one tensor) and join_bn_relu (concatenation)
        c0 = ops.sync_stats_collectives()
        Ck, Cv = 6, 10
        bk, bv = sync_bn(Ck, 2), sync_bn(Cv, 3)
        x = xg[sl].float().to(dev).requires_grad_(True)
        a, bb = ops.split_bn(x, bk, bv)
        j = ops.join_bn_relu([a, bb], [sync_bn(Ck, 4), sync_bn(Cv, 5)])
        (j * cg[sl].float().to(dev)).sum().backward()
        xr = xg.clone().requires_grad_(True)
        ar = _ref_bn(xr[:, :Ck], bk.weight.detach().double().cpu(), bk.bias.detach().double().cpu(), bk.eps, False)
        br = _ref_bn(xr[:, Ck:], bv.weight.detach().double().cpu(), bv.bias.detach().double().cpu(), bv.eps, False)
        torch.manual_seed(4)
        t4 = torch.nn.BatchNorm1d(Ck)
        w4, b4 = (torch.rand(Ck) + 0.5).double(), None
        # (rebuild the join norms' parameters exactly as sync_bn(…, 4/5) drew them)
        torch.manual_seed(4); _ = torch.nn.BatchNorm1d(Ck); w4 = (torch.rand(Ck) + 0.5).double(); b4 = (torch.randn(Ck) * 0.1).double()
        torch.manual_seed(5); _ = torch.nn.BatchNorm1d(Cv); w5 = (torch.rand(Cv) + 0.5).double(); b5 = (torch.randn(Cv) * 0.1).double()
        jr = torch.cat([_ref_bn(ar, w4, b4, 1e-5, True), _ref_bn(br, w5, b5, 1e-5, True)], dim=1)
        (jr * cg).sum().backward()
        res["split_join_y"] = float((j.detach().cpu().double() - jr.detach()[sl]).abs().max())
        res["split_join_gx"] = float((x.grad.cpu().double() - xr.grad[sl]).abs().max())
        res["collectives_split_join"] = ops.sync_stats_collectives() - c0

        # ---- 3. union_keys_values: stacked projections + four norms, one exchange each way
        c0 = ops.sync_stats_collectives()
        H, F = 2, 5
        xin = torch.randn(Bg, Cin, N, generator=g, dtype=torch.float64)
        convs, kbs, vbs = [], [], []
        for hi in range(2):
            torch.manual_seed(10 + hi)
            convs.append(torch.nn.Conv1d(Cin, H * (F + 3), 1, bias=False).to(dev))
            kbs.append(sync_bn(H * 3, 20 + hi))
            vbs.append(sync_bn(H * F, 30 + hi))
        x = xin[sl].float().to(dev).requires_grad_(True)
        assert ops.union_keys_values_eligible(x, convs, kbs, vbs)
        outs = ops.union_keys_values(x, convs, kbs, vbs)
        cots = [torch.randn(Bg, o.size(1), N, generator=g, dtype=torch.float64) for pair in outs for o in pair]
        flat = [o for pair in outs for o in pair]
        sum((o * c[sl].float().to(dev)).sum() for o, c in zip(flat, cots)).backward()
        xr = xin.clone().requires_grad_(True)
        refs = []
        for conv, kb, vb in zip(convs, kbs, vbs):
            yv = torch.einsum("oc,bcn->bon", conv.weight.detach().double().cpu()[:, :, 0], xr)
            refs.append(_ref_bn(yv[:, :H * 3], kb.weight.detach().double().cpu(), kb.bias.detach().double().cpu(), kb.eps, False))
            refs.append(_ref_bn(yv[:, H * 3:], vb.weight.detach().double().cpu(), vb.bias.detach().double().cpu(), vb.eps, False))
        sum((o * c).sum() for o, c in zip(refs, cots)).backward()
        res["union_y"] = max(float((o.detach().cpu().double() - r.detach()[sl]).abs().max()) for o, r in zip(flat, refs))
        res["union_gx"] = float((x.grad.cpu().double() - xr.grad[sl]).abs().max() / xr.grad.abs().max())
        res["collectives_union"] = ops.sync_stats_collectives() - c0
        q.put((rank, res))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_fused_norm_groups_exchange_statistics_across_two_ranks():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        r = got[rank]
        for k in ("bn_relu_y", "bn_relu_gx", "split_join_y", "split_join_gx", "union_y", "running_mean", "running_var"):
            assert r[k] <= 2e-5, (rank, k, r[k])
        for k in ("bn_relu_gw", "bn_relu_gb", "union_gx"):
            assert r[k] <= 1e-5, (rank, k, r[k])
        assert r["nbt"] == 1
        # one all_gather forward + one all_reduce backward per norm GROUP, however many norms it holds
        assert r["collectives_bn_relu"] == 2
        assert r["collectives_split_join"] == 4        # split_bn (2 norms) and join_bn_relu (2 norms): 2 groups
        assert r["collectives_union"] == 2             # 4 norms, one group


def test_sync_batchnorm_at_world_size_one_takes_the_fully_fused_kernels():
    """A SyncBatchNorm without a process group (or with a single rank) needs no exchange: same kernels, same values as
    BatchNorm1d."""
    from cloud_transformers_amd import ops
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(8).cuda().train()
    sbn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.BatchNorm1d(8)).cuda().train()
    x = torch.randn(2, 8, 128, device="cuda")
    assert ops.bn_relu_eligible(sbn, x) and ops._sync_group(sbn) is None
    before = ops.sync_stats_collectives()
    assert torch.equal(ops.bn_relu(x, bn), ops.bn_relu(x, sbn))
    assert ops.sync_stats_collectives() == before

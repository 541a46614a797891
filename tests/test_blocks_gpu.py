"""GPU parity of the MHCT nn.Modules against the reference's outputs
(tests/golden/blocks.npz: state dict + inputs + outputs captured from the
reference) and, for gradients, against the CPU oracle's autograd."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _sd(d):
    return {k[3:]: T(v.copy()) for k, v in d.items() if k.startswith("sd/")}


def _build(case):
    from cloud_transformers_amd.layers import multihead_ct as M
    D = 32
    if case in ("mh2d", "mh2d_pad"):
        return M.MultiHead(D, 4, D, 16, 2, 4)
    if case == "mh3d":
        return M.MultiHead(D, 4, D, 8, 3, 2, scales=True)
    if case == "pool":
        return M.MultiHeadPool(D, 4, 8, 3, 2)
    if case == "adain":
        return M.MultiHeadAdaIn(D, 4, D, 16, 2, 4, n_latent=24)
    if case == "union":
        return M.MultiHeadUnion(D, [4, 4], [16, 8], [2, 3], [4, 2])
    if case == "union_proj":
        return M.MultiHeadUnion(D, [4, 4], [16, 8], [2, 3], [4, 2], model_dim_out=48)
    if case == "union_adain":
        return M.MultiHeadUnionAdaIn(D, [4, 4], [16, 8], [2, 3], [4, 2], n_latent=24)
    raise KeyError(case)


def _stats_occ(stats):
    if isinstance(stats, list):
        return np.array([float(s[0]) for s in stats], dtype=np.float32)
    return np.array([float(stats[0])], dtype=np.float32)


@pytest.mark.parametrize("case", ["mh2d", "mh3d", "mh2d_pad", "pool", "adain", "union", "union_proj", "union_adain"])
def test_block_forward_matches_reference(case):
    d = load_golden("blocks")[case]
    m = _build(case)
    for mode in [mm for mm in ("eval", "train") if f"{mm}_out" in d]:
        m.load_state_dict(_sd(d), strict=True)          # the reference's own key names
        m = m.cuda().train(mode == "train")
        x, pcd = T(d["x"]).cuda(), T(d["pcd"]).cuda()
        if "style" in d:
            res, stats = m(x, T(d["style"]).cuda(), pcd)
        elif "pad" in d:
            res, stats = m(x, (pcd, T(d["pad"]).cuda()))
        else:
            res, stats = m(x, pcd)
        ref = d[f"{mode}_out"]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(res.detach().cpu().numpy(), ref, atol=1e-4 * scale, rtol=1e-4)
        np.testing.assert_allclose(_stats_occ(stats), d[f"{mode}_occ"], rtol=1e-6)
        if f"{mode}_mean" in d and not isinstance(stats, list):
            np.testing.assert_allclose(float(stats[1]), d[f"{mode}_mean"][0], atol=1e-5)
            np.testing.assert_allclose(float(stats[2]), d[f"{mode}_var"][0], rtol=1e-4)


def test_multihead_backward_matches_oracle_autograd():
    """Gradients of a whole MultiHead block (train-mode BN) wrt input, xyz and every
    parameter, against autograd through the CPU oracle."""
    d = load_golden("blocks")["mh2d"]
    sd = _sd(d)
    x0, pcd0 = T(d["x"]), T(d["pcd"])
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(d["train_out"].shape, generator=g)

    sdr = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k and "tensor_mod" not in k)
           for k, v in sd.items()}
    xr, pr = x0.clone().requires_grad_(True), pcd0.clone().requires_grad_(True)
    res, _, _, _ = R.multihead(sdr, xr, pr, in_feature_dim=4, tensor_size=16, tensor_dim=2, heads=4, train=True)
    (res * cot).sum().backward()

    m = _build("mh2d")
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    xc, pc = x0.cuda().requires_grad_(True), pcd0.cuda().requires_grad_(True)
    out, _ = m(xc, pc)
    (out * cot.cuda()).sum().backward()

    def close(a, b, name):
        a, b = a.detach().cpu(), b.detach()
        tol = 2e-4 * max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= tol, (name, float((a - b).abs().max()), tol)

    close(xc.grad, xr.grad, "input")
    close(pc.grad, pr.grad, "orig_pcd")
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        close(p.grad, sdr[name].grad, name)


def test_reference_style_zoo_model_runs():
    """A reference-style model file importing the reference's module paths
    (layers.*, unet2d.*) builds and trains one step on the HIP path."""
    import torch.nn as nn
    from layers.multihead_ct import MultiHeadUnion
    from layers.multihead_ct_pool import MultiHeadPool
    from layers.v2v_groups import Pool3DBlock, Res3DBlock
    from unet2d.unet_parts import Res2DBlock

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.first = nn.Sequential(nn.Conv1d(3, 64, 1, bias=False), nn.BatchNorm1d(64), nn.ReLU(inplace=True))
            self.u1 = MultiHeadUnion(64, [4, 4], [32, 8], [2, 3], [8, 8])
            self.u2 = MultiHeadUnion(64, [8, 8], [16, 8], [2, 3], [8, 8])
            self.pool2d = MultiHeadPool(64, 4, 8, 2, 8)
            self.pool3d = MultiHeadPool(64, 4, 8, 3, 8)
            self.c2 = Res2DBlock(32, 64, groups=8)
            self.c3 = nn.Sequential(Res3DBlock(32, 64, groups=8), Pool3DBlock(2))
            self.head = nn.Linear(128, 5)

        def forward(self, x):
            xyz = x[:, :3, 0]
            f = self.first(xyz)
            f, s1 = self.u1(f, xyz)
            f, s2 = self.u2(f, xyz)
            g2, _ = self.pool2d(f, xyz)
            g3, _ = self.pool3d(f, xyz)
            v = torch.cat([self.c2(g2).mean(dim=(2, 3)), self.c3(g3).mean(dim=(2, 3, 4))], dim=1)
            return self.head(v), s1 + s2

    torch.manual_seed(0)
    model = Model().cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = (torch.rand(2, 3, 1, 512, device="cuda") * 2 - 1)
    y = torch.tensor([1, 3], device="cuda")
    losses = []
    for _ in range(3):
        opt.zero_grad()
        logits, stats = model(x)
        loss = nn.functional.cross_entropy(logits, y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert len(stats) == 4 and all(torch.isfinite(s[0]) for s in stats)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_pointwise_conv1d_matches_conv1d():
    """PointwiseConv1d (rocBLAS batched-GEMM weight gradient) against nn.Conv1d evaluated in float64 on the CPU;
    same state-dict keys as the nn.Conv1d it replaces."""
    from cloud_transformers_amd.layers.pointwise import PointwiseConv1d
    torch.manual_seed(11)
    for cin, cout, bias in [(64, 48, False), (33, 17, True)]:
        m = PointwiseConv1d(cin, cout, kernel_size=1, bias=bias)
        assert sorted(m.state_dict()) == sorted(torch.nn.Conv1d(cin, cout, 1, bias=bias).state_dict())
        x = torch.randn(3, cin, 257)
        gy = torch.randn(3, cout, 257)
        xr = x.double().requires_grad_(True)
        wr = m.weight.detach().double().requires_grad_(True)
        br = m.bias.detach().double().requires_grad_(True) if bias else None
        torch.nn.functional.conv1d(xr, wr, br).backward(gy.double())
        m = m.cuda()
        xc = x.cuda().requires_grad_(True)
        y = m(xc)
        y.backward(gy.cuda())
        np.testing.assert_allclose(xc.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(m.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol=2e-4)
        if bias:
            np.testing.assert_allclose(m.bias.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=2e-4)
    # anything that is not a plain 1x1 projection falls through to nn.Conv1d
    k3 = PointwiseConv1d(8, 8, kernel_size=3, padding=1).cuda()
    assert k3(torch.randn(1, 8, 16, device="cuda")).shape == (1, 8, 16)


@pytest.mark.parametrize("kind", ["bn", "adain"])
def test_union_fusions_equal_the_per_head_path(kind):
    """The union blocks' stacked keys_values_pred GEMM, norms on channel ranges and heads writing into the concatenation
    against the same block evaluated head by head (fusions disabled): output and every gradient."""
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers import multihead_ct as M
    torch.manual_seed(12)
    B, D, N = 2, 32, 512
    if kind == "bn":
        blk = M.MultiHeadUnion(D, [4, 4], [16, 8], [2, 3], [4, 2]).cuda().train()
    else:
        blk = M.MultiHeadUnionAdaIn(D, [4, 4], [16, 8], [2, 3], [4, 2], n_latent=24).cuda().train()
        with torch.no_grad():
            for a in blk.attentions:
                a.scale.fill_(0.3)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if "key_bn.weight" in n:
                p.fill_(0.2)
    x0 = torch.randn(B, D, N, device="cuda")
    pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1
    style = torch.randn(B, 24, device="cuda")
    cot = torch.randn(B, D, N, device="cuda")

    def run():
        blk.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out, _ = blk(x, pcd) if kind == "bn" else blk(x, style, pcd)
        (out * cot).sum().backward()
        return out.detach(), x.grad, {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}

    fused = run()
    saved = (ops.union_keys_values_eligible, ops.bn_relu_eligible, M.MultiHeadUnionAdaIn._fused_keys_values, M.AdaIn1dUpd)
    ops.union_keys_values_eligible = lambda *a, **k: False
    ops.bn_relu_eligible = lambda *a, **k: False
    M.MultiHeadUnionAdaIn._fused_keys_values = lambda self, x, style, **k: None
    M.AdaIn1dUpd = type("NotAdaIn", (), {})           # `type(n[0]) is AdaIn1dUpd` fails: the heads' after stacks run one by one
    try:
        plain = run()
    finally:
        ops.union_keys_values_eligible, ops.bn_relu_eligible, M.MultiHeadUnionAdaIn._fused_keys_values, M.AdaIn1dUpd = saved
    assert torch.allclose(fused[0], plain[0], rtol=1e-4, atol=1e-4)
    assert torch.allclose(fused[1], plain[1], rtol=1e-3, atol=1e-3 * max(1.0, float(plain[1].abs().max())))
    assert fused[2].keys() == plain[2].keys()
    for n in plain[2]:
        a, b = fused[2][n], plain[2][n]
        assert float((a - b).abs().max()) <= 2e-3 * max(1.0, float(b.abs().max())), (n, float((a - b).abs().max()))


@pytest.mark.parametrize("kind", ["bn", "adain"])
def test_heads_on_side_streams_equal_the_serial_block(kind, monkeypatch):
    """CLOUDCT_HEAD_STREAMS=1 (layers.multihead_ct._run_heads): the heads' chains on forked streams — eager, and replayed from a
    HIP graph that captured the fork / join — give the serial block's output and gradients (same kernels, same order per head)."""
    from cloud_transformers_amd.layers import multihead_ct as M
    torch.manual_seed(21)
    B, D, N = 2, 32, 1024
    if kind == "bn":
        blk = M.MultiHeadUnion(D, [4, 4], [16, 8], [2, 3], [4, 2]).cuda().train()
    else:
        blk = M.MultiHeadUnionAdaIn(D, [4, 4], [16, 8], [2, 3], [4, 2], n_latent=24).cuda().train()
    x0 = torch.randn(B, D, N, device="cuda")
    pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1
    style = torch.randn(B, 24, device="cuda")
    cot = torch.randn(B, D, N, device="cuda")
    state = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(x):
        out, _ = blk(x, pcd) if kind == "bn" else blk(x, style, pcd)
        (out * cot).sum().backward()
        return out

    def fresh():
        blk.load_state_dict(state)                     # running statistics back to the start
        blk.zero_grad(set_to_none=True)
        return x0.clone().requires_grad_(True)

    x = fresh()
    want = (run(x).detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None})
    monkeypatch.setattr(M, "HEAD_STREAMS", True)
    x = fresh()
    got = (run(x).detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None})
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0])
    # float-atomic sums in the raster backward differ in the last bits between runs, with or without streams
    np.testing.assert_allclose(got[1].cpu().numpy(), want[1].cpu().numpy(), rtol=1e-4, atol=1e-5)
    for n in want[2]:
        np.testing.assert_allclose(got[2][n].cpu().numpy(), want[2][n].cpu().numpy(), rtol=1e-3, atol=1e-4, err_msg=n)

    # the same through a captured graph
    xs = x0.clone().requires_grad_(True)
    blk.load_state_dict(state)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            blk.zero_grad(set_to_none=True)
            xs.grad = None
            run(xs)
    torch.cuda.current_stream().wait_stream(side)
    blk.load_state_dict(state)
    blk.zero_grad(set_to_none=True)
    xs.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_s = run(xs)
    blk.load_state_dict(state)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_s.detach(), want[0])
    np.testing.assert_allclose(xs.grad.cpu().numpy(), want[1].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_stacked_style_projection_equals_the_norms_own_linears(monkeypatch):
    """MultiHeadUnionAdaIn with ops.StyleProjFn (one stacked product for the style projections of all its AdaIN norms, the
    kernels reading their column ranges of the result in place) against the same block with every norm's own Linear:
    output, input / style cotangents and every parameter gradient (the Linear weights' and biases' included)."""
    from cloud_transformers_amd.layers import multihead_ct as M
    torch.manual_seed(31)
    B, D, N, L = 3, 32, 512, 24
    blk = M.MultiHeadUnionAdaIn(D, [4, 4], [16, 8], [2, 3], [4, 2], model_dim_out=48, n_latent=L).cuda().train()
    x0 = torch.randn(B, D, N, device="cuda")
    pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1
    s0 = torch.randn(B, L, device="cuda")
    cot = torch.randn(B, 48, N, device="cuda")

    def run():
        blk.zero_grad(set_to_none=True)
        x, st = x0.clone().requires_grad_(True), s0.clone().requires_grad_(True)
        out, _ = blk(x, st, pcd)
        (out * cot).sum().backward()
        return out.detach(), x.grad, st.grad, {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}

    got = run()
    assert any("linear.weight" in n for n in got[3])
    monkeypatch.setattr(M.MultiHeadUnionAdaIn, "_style_norms", lambda self, style: [])
    want = run()
    assert got[3].keys() == want[3].keys()
    np.testing.assert_allclose(got[0].cpu().numpy(), want[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[1].cpu().numpy(), want[1].cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(got[2].cpu().numpy(), want[2].cpu().numpy(), rtol=1e-4, atol=1e-4)
    for n in want[3]:
        if n.endswith("conv.0.bias"):
            continue        # a bias in front of an instance norm: its true gradient is 0, both runs hold rounding noise of ~1e-4
        np.testing.assert_allclose(got[3][n].cpu().numpy(), want[3][n].cpu().numpy(), rtol=1e-3, atol=1e-4, err_msg=n)


def test_unstacked_heads_on_side_streams_capture(monkeypatch):
    """The heads' own keys_values_pred projections (union fusion off) inside the forked chains of a captured graph: the pointwise
    GEMMs' backward must not fork a second level of streams there (ops.forked_streams) — capture ends cleanly and the replay
    gives the eager gradients."""
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers import multihead_ct as M
    torch.manual_seed(23)
    B, D, N = 2, 128, 1024
    blk = M.MultiHeadUnion(D, [8, 8], [16, 8], [2, 3], [16, 16]).cuda().train()
    monkeypatch.setattr(ops, "union_keys_values_eligible", lambda *a, **k: False)
    x0 = torch.randn(B, D, N, device="cuda")
    pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1
    cot = torch.randn(B, D, N, device="cuda")
    state = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(x):
        out, _ = blk(x, pcd)
        (out * cot).sum().backward()
        return out

    x = x0.clone().requires_grad_(True)
    want = (run(x).detach().clone(), x.grad.clone())
    xs = x0.clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            blk.load_state_dict(state)
            blk.zero_grad(set_to_none=True)
            xs.grad = None
            run(xs)
    torch.cuda.current_stream().wait_stream(side)
    blk.load_state_dict(state)
    blk.zero_grad(set_to_none=True)
    xs.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_s = run(xs)
    blk.load_state_dict(state)
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_s.detach().cpu().numpy(), want[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xs.grad.cpu().numpy(), want[1].cpu().numpy(), rtol=1e-4, atol=1e-5)

"""GPU parity of the fused training-mode BatchNorm1d + ReLU (ct_bn_relu_fwd / _bwd) against torch's own
F.batch_norm + relu evaluated in float64 on the CPU: output, running statistics, and all three gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, b, rm, rv, eps, momentum, relu, gy):
    x = x.double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    b = b.double().requires_grad_(True)
    rm, rv = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(x, rm, rv, w, b, True, momentum, eps)
    if relu:
        y = torch.relu(y)
    y.backward(gy.double())
    return y.detach(), rm, rv, x.grad, w.grad, b.grad


@pytest.mark.parametrize("B,C,N,relu", [(8, 64, 4096, True), (8, 48, 2048, True), (2, 5, 256, False), (3, 7, 1000, True),
                                        (1, 3, 4, True), (8, 16, 4096, False), (16, 24, 4096, True), (3, 9, 1001, True),
                                        (2, 4, 40000, False)])
def test_bn_relu_matches_torch(B, C, N, relu):
    from cloud_transformers_amd import ops
    torch.manual_seed(B * 100 + C)
    bn = torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.1)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    x = torch.randn(B, C, N) * 3 + 0.7
    gy = torch.randn(B, C, N)
    y_ref, rm_ref, rv_ref, gx_ref, gw_ref, gb_ref = _ref(x, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                        bn.eps, bn.momentum, relu, gy)
    bn = bn.cuda().train()
    xc = x.cuda().requires_grad_(True)
    assert ops.bn_relu_eligible(bn, xc)
    y = ops.bn_relu(xc, bn, relu=relu)
    y.backward(gy.cuda())
    tol = dict(rtol=1e-4, atol=1e-4)
    assert torch.allclose(y.detach().cpu().double(), y_ref, **tol)
    assert torch.allclose(bn.running_mean.cpu().double(), rm_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), rv_ref, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    scale = max(1.0, float(gx_ref.abs().max()))
    assert float((xc.grad.cpu().double() - gx_ref).abs().max()) <= 1e-4 * scale
    assert torch.allclose(bn.weight.grad.cpu().double(), gw_ref, rtol=1e-4, atol=1e-3 * max(1.0, float(gw_ref.abs().max()) * 1e-1))
    assert torch.allclose(bn.bias.grad.cpu().double(), gb_ref, rtol=1e-4, atol=1e-3 * max(1.0, float(gb_ref.abs().max()) * 1e-1))


def test_run_after_dispatch():
    """The blocks' `after` stacks: BatchNorm1d + ReLU in training mode runs fused (register-resident, long-channel and
    scalar-row kernels); eval mode and SyncBatchNorm go through the modules' own forward — with the same result."""
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers.multihead_ct import run_after
    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.BatchNorm1d(32), torch.nn.ReLU(inplace=True)).cuda()
    ref = torch.nn.Sequential(torch.nn.BatchNorm1d(32), torch.nn.ReLU()).cuda()
    ref.load_state_dict(seq.state_dict())
    calls = []
    real = ops.bn_relu
    ops.bn_relu = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        x = torch.randn(4, 32, 512, device="cuda")
        out = run_after(seq, x.clone())
        assert calls == [1]
        assert torch.allclose(out, ref(x), rtol=1e-4, atol=1e-5)
        assert torch.allclose(seq[0].running_var, ref[0].running_var, rtol=1e-5, atol=1e-6)
        del calls[:]
        seq.eval(), ref.eval()
        assert torch.allclose(run_after(seq, x.clone()), ref(x), rtol=1e-5, atol=1e-6) and not calls
        seq.train(), ref.train()
        odd = x[:, :, :511].contiguous()                             # rows that are not float4-addressable: scalar loop kernels
        assert torch.allclose(run_after(seq, odd.clone()), ref(odd), rtol=1e-4, atol=1e-5) and calls == [1]
        del calls[:]
        big = torch.randn(16, 32, 4096, device="cuda")              # B*N > 32768: the channel is re-read per pass
        assert torch.allclose(run_after(seq, big.clone()), ref(big), rtol=1e-4, atol=1e-5) and calls == [1]
        assert torch.allclose(seq[0].running_var, ref[0].running_var, rtol=1e-5, atol=1e-6)
        del calls[:]
        # SyncBatchNorm (what DDP training converts every norm into) takes the fused kernels too: without a process group
        # there is nothing to exchange (tests/test_syncbn_gpu.py covers the two-rank exchange)
        sync = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(torch.nn.BatchNorm1d(32), torch.nn.ReLU())).cuda()
        assert ops.bn_relu_eligible(sync[0], x)
        assert torch.allclose(run_after(sync, x.clone()), ref(x), rtol=1e-4, atol=1e-5) and calls == [1]
    finally:
        ops.bn_relu = real


def test_abi_argument_checks():
    from cloud_transformers_amd import _lib
    lib = _lib.load()
    assert lib.ct_bn_relu_supported(8, 512, 4096) == 1
    assert lib.ct_bn_relu_supported(16, 512, 4096) == 1          # long channels: loop kernels
    assert lib.ct_bn_relu_supported(8, 512, 4095) == 1           # rows that are not float4-addressable: scalar loops
    assert lib.ct_bn_relu_supported(1, 4, 1) == 0                # one value per channel has no variance
    buf = torch.zeros(1 << 16, device="cuda")
    p = buf.data_ptr()
    assert lib.ct_bn_relu_fwd(p, 0, p, p, p, None, None, None, 0, p, 0, p, p, 2, 8, 64, 1e-5, 0.1, 1, None) == -1           # one running buffer only
    assert lib.ct_bn_relu_fwd(p, 8 * 64 - 4, p, p, None, None, None, None, 0, p, 0, p, p, 2, 8, 64, 1e-5, 0.1, 1, None) == -1  # batch stride < C*N
    assert lib.ct_bn_relu_fwd(p, 0, p, p, None, None, None, None, 0, p, 0, p, p, 1, 8, 1, 1e-5, 0.1, 1, None) == -1         # B*N < 2
    assert lib.ct_bn_relu_fwd(None, 0, p, p, None, None, None, None, 0, p, 0, p, p, 2, 8, 64, 1e-5, 0.1, 1, None) == -1


def test_split_bn_equals_the_two_modules_on_split_views():
    """key_bn / values_bn on the halves of one tensor (layers/multihead_ct.py:89-91): values, running statistics and
    every gradient equal torch's BatchNorm1d modules applied to torch.split views."""
    from cloud_transformers_amd import ops
    torch.manual_seed(3)
    B, Ck, Cv, N = 4, 12, 40, 1024
    mods = [torch.nn.BatchNorm1d(Ck), torch.nn.BatchNorm1d(Cv)]
    with torch.no_grad():
        for m in mods:
            m.weight.uniform_(0.5, 1.5)
            m.bias.uniform_(-0.5, 0.5)
    ref = [torch.nn.BatchNorm1d(Ck), torch.nn.BatchNorm1d(Cv)]
    for r, m in zip(ref, mods):
        r.load_state_dict(m.state_dict())
        r.double()
    x = torch.randn(B, Ck + Cv, N) * 2 + 0.3
    gk, gv = torch.randn(B, Ck, N), torch.randn(B, Cv, N)
    xr = x.double().requires_grad_(True)
    a, b = torch.split(xr, [Ck, Cv], dim=1)
    ya, yb = ref[0](a), ref[1](b)
    (ya * gk.double()).sum().backward(retain_graph=True)
    (yb * gv.double()).sum().backward()
    mods = [m.cuda().train() for m in mods]
    xc = x.cuda().requires_grad_(True)
    assert ops.bn_relu_eligible(mods[0], xc, Ck) and ops.bn_relu_eligible(mods[1], xc, Cv)
    ka, kb = ops.split_bn(xc, mods[0], mods[1])
    ((ka * gk.cuda()).sum() + (kb * gv.cuda()).sum()).backward()
    tol = dict(rtol=1e-4, atol=1e-4)
    assert torch.allclose(ka.detach().cpu().double(), ya.detach(), **tol) and torch.allclose(kb.detach().cpu().double(), yb.detach(), **tol)
    assert torch.allclose(xc.grad.cpu().double(), xr.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(xr.grad.abs().max())))
    for m, r in zip(mods, ref):
        assert torch.allclose(m.running_mean.cpu().double(), r.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(m.running_var.cpu().double(), r.running_var, rtol=1e-5, atol=1e-6)
        assert int(m.num_batches_tracked) == 1
        assert torch.allclose(m.weight.grad.cpu().double(), r.weight.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(m.bias.grad.cpu().double(), r.bias.grad, rtol=1e-4, atol=1e-3)
    # only one output used downstream: the other half's cotangent is None -> zeros
    xc2 = x.cuda().requires_grad_(True)
    ka2, _ = ops.split_bn(xc2, mods[0], mods[1])
    ka2.sum().backward()
    assert float(xc2.grad[:, Ck:].abs().max()) == 0.0


def test_residual_and_strided_cotangent():
    """run_after(after, x, residual): the skip connection is added inside the kernel and receives the output cotangent;
    a cotangent that is a channel slice of a wider tensor (torch.cat's backward) is read where it lies."""
    from cloud_transformers_amd.layers.multihead_ct import run_after
    torch.manual_seed(9)
    B, C, N = 4, 24, 512
    seq = torch.nn.Sequential(torch.nn.BatchNorm1d(C), torch.nn.ReLU(inplace=True)).cuda()
    ref = torch.nn.Sequential(torch.nn.BatchNorm1d(C), torch.nn.ReLU()).cuda()
    with torch.no_grad():
        seq[0].weight.uniform_(0.5, 1.5)
        seq[0].bias.uniform_(-0.5, 0.5)
    ref.load_state_dict(seq.state_dict())
    x = torch.randn(B, C, N, device="cuda")
    res = torch.randn(B, C, N, device="cuda")
    wide = torch.randn(B, C + 8, N, device="cuda")
    outs = []
    for mod, fused in ((seq, True), (ref, False)):
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        w = wide.clone().requires_grad_(True)
        y = run_after(mod, xi, ri) if fused else ri + mod(xi)
        z = torch.cat([y, w[:, :8]], dim=1)               # y's cotangent arrives as a slice of z's
        (z * wide).sum().backward()
        outs.append((y.detach(), xi.grad, ri.grad, mod[0].weight.grad.clone(), mod[0].bias.grad.clone(),
                     mod[0].running_mean.clone(), int(mod[0].num_batches_tracked)))
    for a, b in zip(outs[0][:6], outs[1][:6]):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-4), float((a - b).abs().max())
    assert outs[0][6] == outs[1][6] == 1


def test_join_bn_relu_equals_cat_of_the_modules():
    from cloud_transformers_amd import ops
    torch.manual_seed(4)
    B, N, Cs = 4, 512, (16, 40, 8)
    mods = [torch.nn.BatchNorm1d(c).cuda() for c in Cs]
    refs = [torch.nn.BatchNorm1d(c).cuda() for c in Cs]
    for m, r in zip(mods, refs):
        with torch.no_grad():
            m.weight.uniform_(0.5, 1.5)
            m.bias.uniform_(-0.5, 0.5)
        r.load_state_dict(m.state_dict())
    xs = [torch.randn(B, c, N, device="cuda") for c in Cs]
    cot = torch.randn(B, sum(Cs), N, device="cuda")
    xa = [x.clone().requires_grad_(True) for x in xs]
    xb = [x.clone().requires_grad_(True) for x in xs]
    ya = ops.join_bn_relu(xa, mods)
    yb = torch.cat([torch.relu(r(x)) for r, x in zip(refs, xb)], dim=1)
    (ya * cot).sum().backward()
    (yb * cot).sum().backward()
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-5)
    for a, b, m, r in zip(xa, xb, mods, refs):
        assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=2e-4)
        assert torch.allclose(m.weight.grad, r.weight.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(m.bias.grad, r.bias.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(m.running_var, r.running_var, rtol=1e-5, atol=1e-6)
        assert int(m.num_batches_tracked) == int(r.num_batches_tracked) == 1


def test_union_keys_values_equals_per_head_modules():
    """One stacked GEMM + norms for all heads (ops.union_keys_values) against conv_i -> split -> key_bn_i / values_bn_i."""
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers.pointwise import PointwiseConv1d
    torch.manual_seed(8)
    B, Cin, N = 4, 32, 512
    spec = [(12, 16), (6, 40)]                      # (3H, H*C) per head
    def build():
        torch.manual_seed(8)
        convs = [PointwiseConv1d(Cin, ck + cv, kernel_size=1, bias=False).cuda() for ck, cv in spec]
        kbs = [torch.nn.BatchNorm1d(ck).cuda() for ck, _ in spec]
        vbs = [torch.nn.BatchNorm1d(cv).cuda() for _, cv in spec]
        with torch.no_grad():
            for bn in kbs + vbs:
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.5, 0.5)
        return convs, kbs, vbs
    x0 = torch.randn(B, Cin, N, device="cuda")
    cots = [(torch.randn(B, ck, N, device="cuda"), torch.randn(B, cv, N, device="cuda")) for ck, cv in spec]
    res = []
    for fused in (True, False):
        convs, kbs, vbs = build()
        x = x0.clone().requires_grad_(True)
        if fused:
            assert ops.union_keys_values_eligible(x, convs, kbs, vbs)
            outs = ops.union_keys_values(x, convs, kbs, vbs)
        else:
            outs = []
            for conv, kb, vb, (ck, cv) in zip(convs, kbs, vbs, spec):
                a, b = torch.split(conv(x), [ck, cv], dim=1)
                outs.append((kb(a.contiguous()), vb(b.contiguous())))
        sum((k * gk).sum() + (v * gv).sum() for (k, v), (gk, gv) in zip(outs, cots)).backward()
        res.append(([t.detach() for kv in outs for t in kv], x.grad, [c.weight.grad for c in convs],
                    [bn.weight.grad for bn in kbs + vbs], [bn.bias.grad for bn in kbs + vbs],
                    [bn.running_var.clone() for bn in kbs + vbs], [int(bn.num_batches_tracked) for bn in kbs + vbs]))
    f, r = res
    for a, b in zip(f[0], r[0]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
    assert torch.allclose(f[1], r[1], rtol=1e-4, atol=1e-3)
    for group in (2, 3, 4):
        for a, b in zip(f[group], r[group]):
            assert torch.allclose(a, b, rtol=1e-3, atol=2e-3 * max(1.0, float(b.abs().max()))), float((a - b).abs().max())
    for a, b in zip(f[5], r[5]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert f[6] == r[6] == [1, 1, 1, 1]

"""ct_pw_gemm (pointwise-convolution GEMMs, fp32 in / out through split-f16 MFMA terms) against float64 products of the same
operands: the three operand arrangements, ragged sizes, operand magnitudes from 1e-7 to 1e6, exact integer data (layout),
non-finite propagation, determinism of the chunked weight gradient, and the autograd layer on top
(layers/multihead_ct.py:31-33: nn.Conv1d(k=1))."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _needs_split16():
    from cloud_transformers_amd import ops
    if ops.PW_GEMM != "split16":
        pytest.skip("the producers leave operand maxima only for ct_pw_gemm (CLOUDCT_PW_GEMM=split16)")


# relative to sum_k |a_k b_k|: three f16 terms carry 2^-21 of each product; fp32 accumulation adds ~sqrt(K) 2^-24
BOUND = 2.0e-6


def _ref(mode, W, x, gy):
    Wd, xd = W.double(), x.double()
    if mode == 0:
        return torch.matmul(Wd, xd), torch.matmul(Wd.abs(), xd.abs())
    gd = gy.double()
    if mode == 1:
        return torch.matmul(Wd.t(), gd), torch.matmul(Wd.abs().t(), gd.abs())
    return torch.matmul(gd, xd.transpose(1, 2)).sum(0), torch.matmul(gd.abs(), xd.abs().transpose(1, 2)).sum(0)


def _run(mode, W, x, gy):
    from cloud_transformers_amd import ops
    Co, Ci = W.shape
    B, _, N = x.shape
    if mode == 0:
        return ops.pw_gemm(0, W, x, ops.amax(W), ops.amax(x), B, Co, Ci, N)
    if mode == 1:
        return ops.pw_gemm(1, W, gy, ops.amax(W), ops.amax(gy), B, Co, Ci, N)
    return ops.pw_gemm(2, gy, x, ops.amax(gy), ops.amax(x), B, Co, Ci, N)


SHAPES = [(2, 208, 512, 1024), (1, 128, 64, 512), (3, 52, 36, 260), (2, 592, 256, 4096), (1, 4, 4, 4), (5, 132, 260, 36),
          (8, 848, 512, 4096), (2, 512, 64, 16384), (1, 260, 44, 1028), (3, 128, 1056, 256), (1, 516, 520, 8196)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_matches_the_float64_product(shape, mode):
    B, Co, Ci, N = shape
    g = torch.Generator(device="cuda").manual_seed(B * 7 + Co + mode)
    W = torch.randn(Co, Ci, device="cuda", generator=g) / Ci ** 0.5
    x = torch.randn(B, Ci, N, device="cuda", generator=g)
    gy = torch.randn(B, Co, N, device="cuda", generator=g)
    out = _run(mode, W, x, gy)
    ref, mag = _ref(mode, W, x, gy)
    err = (out.double() - ref).abs()
    assert torch.isfinite(out).all()
    assert bool((err <= BOUND * mag + 1e-30).all()), float((err / (mag + 1e-30)).max())


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("scales", [(1e-7, 1.0), (1e6, 1e-5), (3e-4, 2e4), (1.0, 1e-30)])
def test_operand_magnitudes(mode, scales):
    B, Co, Ci, N = 2, 144, 96, 640
    g = torch.Generator(device="cuda").manual_seed(11 + mode)
    W = torch.randn(Co, Ci, device="cuda", generator=g)
    x = torch.randn(B, Ci, N, device="cuda", generator=g)
    gy = torch.randn(B, Co, N, device="cuda", generator=g)
    # a wide spread inside one tensor as well: some rows 2^-12 of the largest
    x[:, ::3] *= 2.0 ** -12
    gy[:, 1::5] *= 2.0 ** -10
    first, second = (W, x) if mode == 0 else (W, gy) if mode == 1 else (gy, x)
    first *= scales[0]
    second *= scales[1]
    out = _run(mode, W, x, gy)
    ref, mag = _ref(mode, W, x, gy)
    err = (out.double() - ref).abs()
    assert bool((err <= BOUND * mag + 1e-44).all()), float((err / (mag + 1e-44)).max())


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_small_integers_are_exact(mode):
    """Every term is exact on small integers, so any slip in a fragment, swizzle or output map shows as a wrong integer."""
    B, Co, Ci, N = 2, 164, 100, 392
    g = torch.Generator(device="cuda").manual_seed(5)
    W = torch.randint(-8, 9, (Co, Ci), device="cuda", generator=g).float()
    x = torch.randint(-8, 9, (B, Ci, N), device="cuda", generator=g).float()
    gy = torch.randint(-8, 9, (B, Co, N), device="cuda", generator=g).float()
    out = _run(mode, W, x, gy)
    ref, _ = _ref(mode, W, x, gy)
    assert torch.equal(out.double(), ref)


def test_zero_and_non_finite_operands():
    from cloud_transformers_amd import ops
    B, Co, Ci, N = 1, 128, 32, 256
    W = torch.zeros(Co, Ci, device="cuda")
    x = torch.randn(B, Ci, N, device="cuda")
    y = ops.pw_gemm(0, W, x, ops.amax(W), ops.amax(x), B, Co, Ci, N)
    assert torch.equal(y, torch.zeros_like(y))
    W = torch.randn(Co, Ci, device="cuda")
    x[0, 3, 17] = float("nan")
    x[0, 5, 100] = float("inf")
    y = ops.pw_gemm(0, W, x, ops.amax(W), ops.amax(x), B, Co, Ci, N)
    assert torch.isnan(y[0, :, 17]).all() and not torch.isfinite(y[0, :, 100]).any()
    keep = torch.ones(N, dtype=torch.bool, device="cuda")
    keep[17] = keep[100] = False
    assert torch.isfinite(y[0][:, keep]).all()


def test_weight_gradient_is_deterministic():
    B, Co, Ci, N = 8, 208, 512, 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, Ci, N, device="cuda", generator=g)
    gy = torch.randn(B, Co, N, device="cuda", generator=g)
    W = torch.empty(Co, Ci, device="cuda")
    a = _run(2, W, x, gy)
    for _ in range(3):
        assert torch.equal(a, _run(2, W, x, gy))


def test_rejects_what_it_does_not_take():
    from cloud_transformers_amd import ops
    x = torch.randn(1, 6, 10, device="cuda")
    W = torch.randn(8, 6, device="cuda")
    with pytest.raises(RuntimeError):
        ops.pw_gemm(0, W, x, None, None, 1, 8, 6, 10)
    assert not ops.pw_eligible(8, 6, 10)


@pytest.mark.parametrize("bias", [False, True])
def test_pointwise_layer_against_float64_conv1d(bias):
    from cloud_transformers_amd.layers.pointwise import PointwiseConv1d
    torch.manual_seed(0)
    B, Ci, Co, N = 4, 192, 336, 2048
    layer = PointwiseConv1d(Ci, Co, 1, bias=bias).cuda()
    ref = torch.nn.Conv1d(Ci, Co, 1, bias=bias).cuda().double()
    ref.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    x = torch.randn(B, Ci, N, device="cuda", requires_grad=True)
    xd = x.detach().double().requires_grad_(True)
    cot = torch.randn(B, Co, N, device="cuda")
    layer(x).backward(cot)
    yd = ref(xd)
    yd.backward(cot.double())
    y = layer(x)
    assert float((y.double() - yd).abs().max()) <= 2e-6 * float(yd.abs().max()) * Ci ** 0.5
    for got, want in [(x.grad, xd.grad), (layer.weight.grad, ref.weight.grad)] + ([(layer.bias.grad, ref.bias.grad)] if bias else []):
        assert float((got.double() - want).abs().max()) <= 3e-6 * float(want.abs().max()) + 1e-30


def test_batchnorm_kernels_leave_the_channel_maxima_of_what_they_write():
    """ct_bn_relu_fwd_amax / _bwd_amax: max |y| (after ReLU and skip) and max |g_x| per channel, bit for bit the maxima of the
    tensors they wrote — the operand scale of the pointwise GEMM that reads them next, without a pass over them."""
    _needs_split16()
    from cloud_transformers_amd import ops
    torch.manual_seed(1)
    for (B, C, N, relu, res) in [(4, 96, 1024, True, True), (2, 40, 260, False, False), (8, 16, 8192, True, False)]:
        bn = torch.nn.BatchNorm1d(C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 2.0)
            bn.bias.uniform_(-1.0, 1.0)
        x = (torch.randn(B, C, N, device="cuda") * 3).requires_grad_(True)
        skip = torch.randn(B, C, N, device="cuda") if res else None
        y = ops.bn_relu(x, bn, relu=relu, residual=skip)
        slots, version = y._ct_amax               # kept 2-D [1, C]: per-ROW maxima for ct_pw_gemm_rs
        assert slots.shape == (1, C)
        assert version == y._version and torch.equal(slots[0], y.detach().abs().amax(dim=(0, 2)))
        assert ops.amax_of(y) is slots
        cot = torch.randn_like(y) * 1e-3
        seen = []
        x.register_hook(lambda g: seen.append(g))
        y.backward(cot)
        gslots, _ = seen[0]._ct_amax
        assert torch.equal(gslots[0], seen[0].abs().amax(dim=(0, 2)))
        y.detach().mul_(2.0)                       # an in-place change: the remembered maxima no longer describe y
        assert ops.amax_of(y) is not slots and float(ops.amax_of(y).max()) == float(y.abs().max())


def test_block_level_convs_reuse_the_producers_maxima(monkeypatch):
    """BatchNorm+ReLU -> PointwiseConv1d -> BatchNorm+ReLU, forward and backward: no ct_amax_f32 pass is left (x and g_y come
    with their producers' maxima, the weight's come out of ct_pw_prep_weight with its transpose); results equal the ones computed
    with a pass over every operand."""
    _needs_split16()
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers.pointwise import PointwiseConv1d
    torch.manual_seed(2)
    B, C, N = 4, 128, 2048
    bn0, bn1 = torch.nn.BatchNorm1d(C).cuda(), torch.nn.BatchNorm1d(256).cuda()
    conv = PointwiseConv1d(C, 256, 1, bias=False).cuda()
    x = torch.randn(B, C, N, device="cuda", requires_grad=True)
    cot = torch.randn(B, 256, N, device="cuda")

    def run():
        for p in list(conv.parameters()) + [x]:
            p.grad = None
        out = ops.bn_relu(conv(ops.bn_relu(x, bn0, relu=True)), bn1, relu=True)
        out.backward(cot)
        return out.detach().clone(), x.grad.clone(), conv.weight.grad.clone()

    calls = []
    real = ops.amax
    monkeypatch.setattr(ops, "amax", lambda t: (calls.append(tuple(t.shape)), real(t))[1])
    got = run()
    assert calls == [], calls
    monkeypatch.setattr(ops, "amax_of", lambda t, rows=False: real(t))
    want = run()
    for a, b in zip(got, want):
        # (per-row scales from the producers' per-channel maxima against one scale per tensor: the same products to rounding)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_adain_kernels_leave_the_row_maxima_of_what_they_write():
    _needs_split16()
    from cloud_transformers_amd import ops
    torch.manual_seed(3)
    for (B, C, N, relu, res) in [(2, 96, 4096, True, True), (3, 40, 260, False, False)]:
        x = (torch.randn(B, C, N, device="cuda") * 2).requires_grad_(True)
        gb = torch.randn(B, 2, C, device="cuda") * 0.5
        skip = torch.randn(B, C, N, device="cuda") if res else None
        y = ops.adain(x, gb, relu=relu, residual=skip)
        slots, _ = y._ct_amax
        assert torch.equal(slots.view(B, C), y.detach().abs().amax(dim=2))
        seen = []
        x.register_hook(lambda g: seen.append(g))
        y.backward(torch.randn_like(y) * 1e-2)
        gslots, _ = seen[0]._ct_amax
        assert torch.equal(gslots.view(B, C), seen[0].abs().amax(dim=2))


def test_random_shapes_all_arrangements():
    """Forty random (B, Co, Ci, N), every dimension a multiple of 4 up to a few tiles / K-steps / chunks, all three arrangements,
    operands with their own magnitudes: the elementwise bound against the float64 product."""
    import random
    rnd = random.Random(1234)
    for case in range(40):
        B = rnd.choice([1, 1, 2, 3, 5, 8])
        Co = 4 * rnd.randint(1, 90)
        Ci = 4 * rnd.randint(1, 90)
        N = 4 * rnd.randint(1, 700)
        g = torch.Generator(device="cuda").manual_seed(case)
        W = torch.randn(Co, Ci, device="cuda", generator=g) * 10.0 ** rnd.uniform(-3, 2)
        x = torch.randn(B, Ci, N, device="cuda", generator=g) * 10.0 ** rnd.uniform(-3, 3)
        gy = torch.randn(B, Co, N, device="cuda", generator=g) * 10.0 ** rnd.uniform(-6, 1)
        for mode in (0, 1, 2):
            out = _run(mode, W, x, gy)
            ref, mag = _ref(mode, W, x, gy)
            err = (out.double() - ref).abs()
            assert bool((err <= BOUND * mag + 1e-30).all()), (case, mode, (B, Co, Ci, N), float((err / (mag + 1e-30)).max()))


def test_syncbn_apply_kernels_leave_the_channel_maxima():
    """The split passes of SyncBatchNorm (statistics -> exchange -> apply; here a 'world' of one rank's buffer fed back in) with
    amax_out: the maxima of what the apply kernels wrote, forward and backward."""
    from cloud_transformers_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(5)
    B, C, N = 4, 48, 1024
    x = torch.randn(B, C, N, device="cuda") * 2
    w, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    local = torch.empty(2 * C + 1, device="cuda")
    _lib.check(lib.ct_bn_stats_fwd(x.data_ptr(), 0, local.data_ptr(), local.data_ptr() + 4 * C, local.data_ptr() + 8 * C, B, C, N, st), "stats")
    y, mean, rstd, count, am = torch.empty_like(x), torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(1, device="cuda"), \
        torch.empty(C, device="cuda")
    _lib.check(lib.ct_bn_apply_fwd_amax(x.data_ptr(), 0, w.data_ptr(), b.data_ptr(), local.data_ptr(), local.data_ptr() + 4 * C,
                                        local.data_ptr() + 8 * C, 1, 2 * C + 1, None, None, None, None, 0, y.data_ptr(), 0, mean.data_ptr(),
                                        rstd.data_ptr(), count.data_ptr(), am.data_ptr(), B, C, N, 1e-5, 0.1, 1, st), "apply_fwd")
    assert torch.equal(am, y.abs().amax(dim=(0, 2)))
    want = torch.relu(torch.nn.functional.batch_norm(x, None, None, w, b, True, 0.1, 1e-5))
    assert float((y - want).abs().max()) < 1e-5
    gy = torch.randn_like(x)
    sums = torch.empty(2 * C, device="cuda")
    _lib.check(lib.ct_bn_reduce_bwd(x.data_ptr(), 0, w.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(), 0,
                                    sums.data_ptr(), sums.data_ptr() + 4 * C, B, C, N, 1, st), "reduce")
    gx, gam = torch.empty_like(x), torch.empty(C, device="cuda")
    _lib.check(lib.ct_bn_apply_bwd_amax(x.data_ptr(), 0, w.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(), 0,
                                        sums.data_ptr(), sums.data_ptr() + 4 * C, count.data_ptr(), gx.data_ptr(), 0, gam.data_ptr(),
                                        B, C, N, 1, st), "apply_bwd")
    assert torch.equal(gam, gx.abs().amax(dim=(0, 2)))


def test_prepared_weight_gives_the_same_data_gradient():
    """ct_pw_prep_weight_rs (row / column maxima + W^T in one launch) and CT_PW_DGRAD_T against ct_amax_f32 + CT_PW_DGRAD: the
    maxima are exact, and with the prepared maxima used as partials of ONE maximum (a flat view: one scale per tensor) the
    two data gradients have the same bits."""
    from cloud_transformers_amd import ops
    torch.manual_seed(9)
    for (B, Co, Ci, N) in [(2, 208, 512, 1024), (3, 52, 36, 260), (1, 640, 132, 512)]:
        W = torch.randn(Co, Ci, device="cuda") * 0.3
        gy = torch.randn(B, Co, N, device="cuda") * 1e-2
        (rowmax, colmax), Wt = ops.prep_weight(W, True)
        assert torch.equal(Wt, W.t().contiguous())
        assert torch.equal(rowmax.amax(0), W.abs().amax(1)) and torch.equal(colmax.amax(0), W.abs().amax(0))
        am_g = ops.amax(gy)
        a = ops.pw_gemm(ops.PW_DGRAD_T, Wt, gy, colmax.reshape(-1), am_g, B, Co, Ci, N)
        b = ops.pw_gemm(ops.PW_DGRAD, W, gy, ops.amax(W), am_g, B, Co, Ci, N)
        assert torch.equal(a, b)


def _run_rs(mode, W, x, gy):
    """The products with per-row maxima wherever the arrangement takes them (ct_pw_gemm_rs through ops.pw_gemm: 2-D maxima):
    forward — rows of W; data gradient — rows of W^T; weight gradient — channels of g_y and of x."""
    from cloud_transformers_amd import ops
    Co, Ci = W.shape
    B, _, N = x.shape
    (rowmax, colmax), Wt = ops.prep_weight(W, True)
    if mode == 0:
        return ops.pw_gemm(0, W, x, rowmax, ops.amax_rows(x), B, Co, Ci, N)
    if mode == 1:
        return ops.pw_gemm(3, Wt, gy, colmax, ops.amax_rows(gy), B, Co, Ci, N)
    return ops.pw_gemm(2, gy, x, ops.amax_rows(gy), ops.amax_rows(x), B, Co, Ci, N)


@pytest.mark.parametrize("depth", [16, 20, 24, 28, 40])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_rows_far_below_the_tensor_maximum_keep_their_precision(mode, depth):
    """Dynamic range INSIDE a tensor (include/cloudct.h, ct_pw_gemm_rs): whole rows of the k-contiguous operands 2^-depth below
    the tensor's maximum — rows of W (forward), columns of W (data gradient), channels of g_y and of x (weight gradient: a
    nearly-dead channel group).  With one scale per tensor such a row sinks into f16's subnormals (2^-24: 1e-5 relative,
    2^-28: nothing left); with the row's own scale every output stays within the fp32-GEMM bound of ITS OWN operands."""
    B, Co, Ci, N = 2, 256, 192, 1024
    g = torch.Generator(device="cuda").manual_seed(17 + mode + depth)
    W = torch.randn(Co, Ci, device="cuda", generator=g) / Ci ** 0.5
    x = torch.randn(B, Ci, N, device="cuda", generator=g)
    gy = torch.randn(B, Co, N, device="cuda", generator=g)
    q = 2.0 ** -depth
    if mode == 0:
        W[5::7] *= q                      # output rows 5, 12, ... of y are tiny
    elif mode == 1:
        W[:, 3::5] *= q                   # output rows 3, 8, ... of g_x
    else:
        gy[:, 2::9] *= q                  # rows of g_W
        x[:, 1::11] *= q                  # columns of g_W
    out = _run_rs(mode, W, x, gy)
    ref, mag = _ref(mode, W, x, gy)
    err = (out.double() - ref).abs()
    assert torch.isfinite(out).all()
    assert bool((err <= BOUND * mag + 1e-44).all()), float((err / (mag + 1e-44)).max())
    # ... and the per-tensor scale does lose them from 2^-24 on (what the bound in the header says; not a requirement)
    if depth >= 28 and depth < 40:
        flat = _run(mode if mode != 1 else 1, W, x, gy)
        bad = float(((flat.double() - ref).abs() / (mag + 1e-44)).max())
        assert bad > BOUND, bad


def test_row_maxima_of_the_producers_reach_the_weight_gradient():
    """The maxima the norm kernels leave are per channel: tag_amax keeps them 2-D and ops.pw_backward hands them to
    ct_pw_gemm_rs as per-row maxima — a block whose cotangent has a nearly-dead channel group gets an exact weight gradient
    through the autograd layer, without any extra pass over the tensors."""
    _needs_split16()
    from cloud_transformers_amd import ops
    B, Co, Ci, N = 2, 256, 128, 2048
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.randn(B, Ci, N, device="cuda", generator=g)
    gy = torch.randn(B, Co, N, device="cuda", generator=g)
    gy[:, 64:96] *= 2.0 ** -26
    W = torch.randn(Co, Ci, device="cuda", generator=g) / Ci ** 0.5
    # as a producer would: per-channel maxima in a slots buffer, tagged on the tensors
    sx, sg = torch.empty(Ci, device="cuda"), torch.empty(Co, device="cuda")
    sx.copy_(x.abs().amax(dim=(0, 2)))
    sg.copy_(gy.abs().amax(dim=(0, 2)))
    ops.tag_amax(x, sx)
    ops.tag_amax(gy, sg)
    assert ops.amax_of(x).shape == (1, Ci) and ops.amax_of(gy).shape == (1, Co)
    y, am_w, am_x, Wt = ops.pw_forward(W, x, True)
    g_x, g_w = ops.pw_backward(W, x, gy, am_w, am_x, True, True, Wt=Wt)
    ref, mag = _ref(2, W, x, gy)
    err = (g_w.double() - ref).abs()
    assert bool((err <= BOUND * mag + 1e-44).all()), float((err / (mag + 1e-44)).max())
    refx, magx = _ref(1, W, x, gy)
    assert bool(((g_x.double() - refx).abs() <= BOUND * magx + 1e-44).all())


@pytest.mark.parametrize("shape", [(2, 208, 512, 1024), (3, 52, 36, 260), (8, 848, 512, 4096), (1, 260, 44, 1028), (1, 4, 4, 4)])
@pytest.mark.parametrize("mode", [0, 1, 3])
def test_addend_in_the_epilogue_is_the_product_plus_the_addend_bit_for_bit(shape, mode):
    """ct_pw_gemm_rs_add (out = product + addend, the addend read in the kernel's epilogue): exactly fl(product + addend) of the
    SAME kernel's product, for the forward and both data-gradient arrangements; the weight gradient refuses an addend."""
    from cloud_transformers_amd import _lib, ops
    B, Co, Ci, N = shape
    g = torch.Generator(device="cuda").manual_seed(B * 11 + Co + mode)
    W = torch.randn(Co, Ci, device="cuda", generator=g) / Ci ** 0.5
    x = torch.randn(B, Ci, N, device="cuda", generator=g)
    gy = torch.randn(B, Co, N, device="cuda", generator=g)
    if mode == 0:
        a, b, am_a, am_b = W, x, ops.amax(W), ops.amax(x)
    elif mode == 1:
        a, b, am_a, am_b = W, gy, ops.amax(W), ops.amax(gy)
    else:
        (rowmax, colmax), Wt = ops.prep_weight(W, True)
        if Wt is None:
            pytest.skip("no prepared transpose for this weight")
        a, b, am_a, am_b = Wt, gy, colmax, ops.amax(gy)
    plain = ops.pw_gemm(mode, a, b, am_a, am_b, B, Co, Ci, N)
    add = torch.randn(plain.shape, device="cuda", generator=g) * 3
    fused = ops.pw_gemm(mode, a, b, am_a, am_b, B, Co, Ci, N, addend=add)
    assert torch.equal(fused, plain + add)
    lib = _lib.load()
    p = ops._ptr
    ws = torch.empty(max(lib.ct_pw_gemm_workspace_bytes(2, B, Co, Ci, N), 16), device="cuda", dtype=torch.uint8)
    gw = torch.empty(Co, Ci, device="cuda")
    assert lib.ct_pw_gemm_rs_add(2, p(gy), p(x), p(gw), p(gw), None, 0, 0, None, 0, 0, p(ws), ws.numel(), B, Co, Ci, N, None) == -1      # CT_EINVAL
    assert lib.ct_pw_gemm_rs_add(0, p(W), p(x), p(plain), p(plain), None, 0, 0, None, 0, 0, None, 0, B, Co, Ci, N, None) == -1   # addend aliases out


def test_union_block_with_the_shortcut_summed_in_the_data_gradient(monkeypatch):
    """MultiHeadUnion (identity shortcut): the block's input leaves the stacked projections' node as an output too and the
    shortcut's cotangent is added in the data gradient's epilogue (ops.UnionKeysValuesFn passthrough) — outputs and every
    gradient equal the path where autograd sums the two cotangents, the input's gradient bit for bit."""
    from cloud_transformers_amd.layers import multihead_ct as M
    from cloud_transformers_amd.layers.pointwise import convert_pointwise
    res = {}
    for flag in (False, True):
        monkeypatch.setattr(M, "SKIP_IN_DGRAD", flag)
        torch.manual_seed(3)
        blk = convert_pointwise(M.MultiHeadUnion(model_dim=128, features_dims=[4, 8], heads=[4, 4], tensor_sizes=[16, 8],
                                                  model_dim_out=128, tensor_dims=[2, 3]).cuda()).train()
        x = torch.randn(2, 128, 1024, device="cuda", requires_grad=True)
        pcd = torch.rand(2, 3, 1024, device="cuda") * 2 - 1
        y, _ = blk(x, pcd)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").reshape(y.shape)).sum().backward()
        res[flag] = (y.detach(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    assert torch.equal(res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1])
    for a, b in zip(res[False][2], res[True][2]):
        assert torch.equal(a, b)


def test_adain_union_block_with_the_shortcut_summed_in_the_data_gradient(monkeypatch):
    """MultiHeadUnionAdaIn, same as above: identical outputs and gradients (input, style, parameters) either way."""
    from cloud_transformers_amd.layers import multihead_ct as M
    from cloud_transformers_amd.layers.pointwise import convert_pointwise
    res = {}
    for flag in (False, True):
        monkeypatch.setattr(M, "SKIP_IN_DGRAD", flag)
        torch.manual_seed(5)
        blk = convert_pointwise(M.MultiHeadUnionAdaIn(model_dim=128, features_dims=[4, 8], heads=[4, 4], tensor_sizes=[16, 8],
                                                      model_dim_out=128, tensor_dims=[2, 3], n_latent=64).cuda()).train()
        x = torch.randn(2, 128, 1024, device="cuda", requires_grad=True)
        style = torch.randn(2, 64, device="cuda", requires_grad=True)
        pcd = torch.rand(2, 3, 1024, device="cuda") * 2 - 1
        y, _ = blk(x, style, pcd)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").reshape(y.shape)).sum().backward()
        res[flag] = (y.detach(), x.grad.clone(), style.grad.clone(), [p.grad.clone() for p in blk.parameters() if p.grad is not None])
    assert torch.equal(res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])
    assert len(res[False][3]) == len(res[True][3])
    for a, b in zip(res[False][3], res[True][3]):
        assert torch.equal(a, b)

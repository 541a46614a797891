"""GPU parity of the MFMA grouped convolution (csrc/ct_gconv.hip) against torch's
conv2d/conv3d evaluated in float64 on the CPU (an independent reference of the same op;
tolerance 1e-5 relative: the MFMA is an exact-fp32 k-ordered fma chain)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # B, groups, Cin, Cout, dim, W, bias
    (2, 4, 4, 4, 2, (16, 16), True),          # MultiHead.conv, C=4
    (2, 3, 16, 16, 2, (32, 32), True),        # north-star head: C=16, 32x32
    (1, 2, 4, 4, 2, (128, 128), True),        # zoo 2D W=128
    (1, 2, 16, 32, 2, (20, 12), False),       # Res2DBlock widening, non-square
    (1, 2, 32, 64, 2, (8, 8), False),         # several 16-row output blocks
    (1, 1, 5, 7, 2, (9, 11), True),           # channels not multiples of 4 / 16
    (2, 2, 4, 4, 3, (8, 8, 8), True),         # 3D
    (1, 2, 16, 16, 3, (16, 16, 16), True),    # 3D C=16 (tiled in depth)
    (1, 1, 4, 4, 3, (32, 32, 32), True),      # zoo 3D W=32
    (1, 2, 32, 32, 3, (8, 8, 8), False),
    (1, 2, 32, 64, 3, (8, 8, 8), False),      # classifier Res3DBlock(512, 1024, groups=16) on the 8^3 pool
    (1, 2, 64, 64, 3, (4, 4, 4), False),      # ... after Pool3DBlock(2): 64 channels per group, filter bank > LDS in the quad form
    (1, 2, 64, 64, 3, (2, 2, 2), False),
    (1, 2, 64, 64, 2, (4, 4), False),         # classifier Res2DBlock(1024, 1024, groups=16) on 4^2
    (1, 3, 16, 48, 2, (8, 8), True),          # few workgroups: the 3 output-channel blocks of a tile go to different workgroups
    (2, 16, 64, 64, 3, (8, 8, 8), False),     # inpainter Res3DBlock(1024, 1024, groups=16) at batch 2
    # wide groups on the K-split kernel (contraction in blocks of 16 input channels)
    (1, 2, 64, 64, 3, (8, 8, 8), True),
    (1, 2, 48, 64, 2, (8, 8), True),          # input channels not a multiple of the block
    (1, 2, 72, 40, 3, (4, 4, 4), True),       # ragged input block and ragged 16-row output block
    (1, 1, 64, 64, 2, (16, 16), False),       # several spans per wave
    # 2^d volumes on the dense small-volume kernel
    (8, 16, 64, 64, 3, (2, 2, 2), True),      # classifier Res3DBlock(1024, 1024, groups=16) after the second pool
    (2, 3, 40, 24, 3, (2, 2, 2), False),      # ragged channel counts, few clouds
    (3, 2, 5, 7, 2, (2, 2), True),
    (70, 1, 8, 8, 3, (2, 2, 2), True),        # more clouds than lanes per output channel
    (1, 2, 160, 16, 2, (2, 2), False),
]


@pytest.mark.parametrize("cfg", CASES, ids=[str(c) for c in CASES])
def test_gconv_fwd_bwd(cfg):
    from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d
    B, G, Cin, Cout, dim, W, bias = cfg
    torch.manual_seed(sum(W) + Cin + Cout)
    cls = GroupedConv3d if dim == 3 else GroupedConv2d
    m = cls(G * Cin, G * Cout, kernel_size=3, stride=1, padding=1, groups=G, bias=bias)
    x = torch.randn(B, G * Cin, *W)
    cot = torch.randn(B, G * Cout, *W)
    # float64 reference on the CPU
    xr = x.double().requires_grad_(True)
    wr = m.weight.detach().double().requires_grad_(True)
    br = m.bias.detach().double().requires_grad_(True) if bias else None
    fn = torch.nn.functional.conv3d if dim == 3 else torch.nn.functional.conv2d
    yr = fn(xr, wr, br, stride=1, padding=1, groups=G)
    (yr * cot.double()).sum().backward()

    m = m.cuda()
    xc = x.cuda().requires_grad_(True)
    from cloud_transformers_amd.layers.gconv import GroupedConvFn
    y = GroupedConvFn.apply(xc, m.weight, m.bias, G)          # the kernels themselves (the module routes wide groups to the library)
    (y * cot.cuda()).sum().backward()

    def close(a, b, name, tol=2e-5):
        a, b = a.detach().cpu().double(), b.detach()
        err = float((a - b).abs().max())
        assert err <= tol * max(1.0, float(b.abs().max())), (name, err)

    close(y, yr, "y")
    close(xc.grad, xr.grad, "g_x")
    close(m.weight.grad, wr.grad, "g_w", 5e-5)
    if bias:
        close(m.bias.grad, br.grad, "g_bias", 5e-5)


def test_wide_groups_run_on_the_k_split_kernels():
    """More than 32 channels per group (the Res2D / Res3D stacks of the classifier and the inpainter on the pooled 8^3 ..
    4^3 volumes and 8^2 .. 4^2 planes) run on this package's kernels too (K-split MFMA forward / backward-data, small-volume
    weight gradient), and so do the 2^3 volumes (dense small-volume kernel); only wide groups on other rows that are not
    16-byte multiples go to the library."""
    from cloud_transformers_amd.layers import gconv as G
    calls = []
    real = G.GroupedConvFn.apply
    G.GroupedConvFn.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
    try:
        narrow = G.GroupedConv3d(64, 64, 3, padding=1, groups=2, bias=False).cuda()       # 32 per group
        wide = G.GroupedConv3d(128, 128, 3, padding=1, groups=2, bias=False).cuda()       # 64 per group
        mixed = G.GroupedConv2d(64, 128, 3, padding=1, groups=2, bias=False).cuda()       # 32 -> 64
        narrow(torch.randn(1, 64, 4, 4, 4, device="cuda"))
        assert calls == [1]
        xw = torch.randn(1, 128, 4, 4, 4, device="cuda")
        assert torch.allclose(wide(xw), torch.nn.functional.conv3d(xw, wide.weight, None, padding=1, groups=2), atol=1e-4)
        assert calls == [1, 1]
        mixed(torch.randn(1, 64, 8, 8, device="cuda"))
        assert calls == [1, 1, 1]
        x2 = torch.randn(1, 128, 2, 2, 2, device="cuda")                                   # 2^3: rows of 2 floats
        assert torch.allclose(wide(x2), torch.nn.functional.conv3d(x2, wide.weight, None, padding=1, groups=2), atol=1e-4)
        assert calls == [1, 1, 1, 1]
        x3 = torch.randn(1, 128, 3, 3, 3, device="cuda")                                   # rows of 3 floats, wide groups
        assert torch.allclose(wide(x3), torch.nn.functional.conv3d(x3, wide.weight, None, padding=1, groups=2), atol=1e-4)
        assert calls == [1, 1, 1, 1]
    finally:
        G.GroupedConvFn.apply = real


@pytest.mark.parametrize("dim,B,G,ci,co,sp,bias", [(2, 1, 2, 4, 4, (5, 5), False), (2, 8, 16, 16, 32, (16, 16), True),
                                                     (3, 8, 16, 32, 64, (8, 8, 8), True), (3, 2, 1, 6, 3, (2, 3, 4), False)])
def test_pointwise_skip_projections_are_batched_gemms(dim, B, G, ci, co, sp, bias):
    """The grouped 1x1 projections of the Res stacks (layers/v2v_groups.py:40-44): forward and all three gradients
    against the stock grouped convolution."""
    from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d
    torch.manual_seed(dim * 100 + G)
    m = (GroupedConv2d if dim == 2 else GroupedConv3d)(G * ci, G * co, kernel_size=1, groups=G, bias=bias).cuda()
    x = torch.randn(B, G * ci, *sp, device="cuda", requires_grad=True)
    go = torch.randn(B, G * co, *sp, device="cuda")
    y = m(x)
    y.backward(go)
    got = [y.detach(), x.grad.clone(), m.weight.grad.clone()] + ([m.bias.grad.clone()] if bias else [])
    xr = x.detach().clone().requires_grad_(True)
    w = m.weight.detach().clone().requires_grad_(True)
    b = m.bias.detach().clone().requires_grad_(True) if bias else None
    conv = torch.nn.functional.conv2d if dim == 2 else torch.nn.functional.conv3d
    yr = conv(xr, w, b, groups=G)
    yr.backward(go)
    want = [yr.detach(), xr.grad, w.grad] + ([b.grad] if bias else [])
    for a, r in zip(got, want):
        assert a.shape == r.shape
        assert torch.allclose(a, r, rtol=1e-4, atol=1e-4 * float(r.abs().max()))


def test_ineligible_configs_use_parent_class():
    from cloud_transformers_amd.layers.gconv import GroupedConv2d
    m = GroupedConv2d(8, 8, kernel_size=3, stride=2, padding=1, groups=2, bias=False).cuda()       # strided
    x = torch.randn(1, 8, 6, 6, device="cuda")
    ref = torch.nn.functional.conv2d(x, m.weight, None, stride=2, padding=1, groups=2)
    assert torch.allclose(m(x), ref, atol=1e-5)


def test_state_dict_is_plain_conv():
    from cloud_transformers_amd.layers.gconv import GroupedConv3d
    m = GroupedConv3d(8, 8, 3, padding=1, groups=2)
    ref = torch.nn.Conv3d(8, 8, 3, padding=1, groups=2)
    assert list(m.state_dict()) == list(ref.state_dict())
    ref.load_state_dict(m.state_dict(), strict=True)


def _rand_cfg(rng):
    dim = int(rng.integers(2, 4))
    rows4 = rng.random() < 0.75                       # most cases on the 16-byte-row kernels (quad / ring / C4 forms)
    Wx = int(rng.choice([4, 8, 12, 16, 20, 32, 64])) if rows4 else int(rng.integers(1, 23))
    H = int(rng.integers(1, 21))
    D = int(rng.integers(1, 13)) if dim == 3 else None
    if rng.random() < 0.3:
        cin = cout = 4                                # vector-ALU kernels
    else:
        cin, cout = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    W = (D, H, Wx) if dim == 3 else (H, Wx)
    return (int(rng.integers(1, 4)), int(rng.integers(1, 4)), cin, cout, dim, W, bool(rng.random() < 0.5))


@pytest.mark.parametrize("seed", range(24))
def test_gconv_fuzz(seed):
    """Random shapes through every kernel family (MFMA one-position and quad forms, four-channel vector-ALU form, ring and
    tile weight gradients, workspace reduction): ragged depth / height, single rows, channel counts off the 4 / 16 grid."""
    from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d
    rng = np.random.default_rng(1000 + seed)
    B, G, Cin, Cout, dim, W, bias = _rand_cfg(rng)
    torch.manual_seed(seed)
    cls = GroupedConv3d if dim == 3 else GroupedConv2d
    m = cls(G * Cin, G * Cout, kernel_size=3, stride=1, padding=1, groups=G, bias=bias)
    x = torch.randn(B, G * Cin, *W)
    cot = torch.randn(B, G * Cout, *W)
    xr = x.double().requires_grad_(True)
    wr = m.weight.detach().double().requires_grad_(True)
    br = m.bias.detach().double().requires_grad_(True) if bias else None
    fn = torch.nn.functional.conv3d if dim == 3 else torch.nn.functional.conv2d
    (fn(xr, wr, br, stride=1, padding=1, groups=G) * cot.double()).sum().backward()
    yr = fn(xr, wr, br, stride=1, padding=1, groups=G)
    m = m.cuda()
    xc = x.cuda().requires_grad_(True)
    from cloud_transformers_amd.layers.gconv import GroupedConvFn
    y = GroupedConvFn.apply(xc, m.weight, m.bias, G)          # the kernels themselves (the module routes wide groups to the library)
    (y * cot.cuda()).sum().backward()

    def close(a, b, name, tol):
        a, b = a.detach().cpu().double(), b.detach()
        err = float((a - b).abs().max())
        assert err <= tol * max(1.0, float(b.abs().max())), (name, (B, G, Cin, Cout, dim, W, bias), err)

    close(y, yr, "y", 2e-5)
    close(xc.grad, xr.grad, "g_x", 2e-5)
    close(m.weight.grad, wr.grad, "g_w", 5e-5)
    if bias:
        close(m.bias.grad, br.grad, "g_bias", 5e-5)


def test_shapes_without_a_tile_plan_take_the_library_convolution():
    """Very wide rows with many channels per group do not fit any LDS tile plan (ct_gconv_supported == 0): the module
    answers through its parent class (MIOpen) instead of failing — found by a 300-seed soak of the fuzz above."""
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.layers.gconv import GroupedConv3d
    lib = _lib.load()
    W = (6, 10, 64)
    assert lib.ct_gconv_supported(2, 2, 38, 22, 3, _lib.int_array(W)) == 0
    assert lib.ct_gconv_supported(8, 16, 16, 16, 3, _lib.int_array((16, 16, 16))) == 1
    torch.manual_seed(0)
    m = GroupedConv3d(2 * 38, 2 * 22, 3, padding=1, groups=2, bias=False).cuda()
    x = torch.randn(2, 2 * 38, *W, device="cuda", requires_grad=True)
    y = m(x)
    y.sum().backward()
    ref = torch.nn.functional.conv3d(x.detach().double(), m.weight.detach().double(), padding=1, groups=2)
    assert float((y.detach().double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert x.grad is not None and m.weight.grad is not None


@pytest.mark.parametrize("B,G,W", [(2, 3, (32, 32, 32)), (1, 3, (5, 7, 16)), (2, 2, (9, 33, 48)), (3, 1, (1, 1, 16)), (2, 5, (2, 20, 64)),
                                   (1, 2, (7, 5, 128)), (8, 16, (32, 32, 32))])
def test_four_channel_3d_groups_on_the_matrix_cores(B, G, W):
    """gconv_c4_mfma3_kernel (three depth taps per MFMA, slices streamed through an LDS ring) forced on — the dispatch takes it
    from ~2 M positions — against float64 and against the vector-ALU kernel: forward and backward-data, ragged depth / height
    (partial row tiles, depth segments), rows of 16 .. 128."""
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.ops import _ptr, _stream
    lib = _lib.load()
    torch.manual_seed(sum(W) + B)
    x = torch.randn(B, G * 4, *W, device="cuda")
    w = torch.randn(G * 4, 4, 3, 3, 3, device="cuda") * 0.1
    b = torch.randn(G * 4, device="cuda")
    Wa = _lib.int_array(W)
    outs = {}
    try:
        for flag in (4, 2):                                   # always / never
            lib.ct_debug_set_gconv(flag)
            y = torch.full_like(x, float("nan"))
            gx = torch.full_like(x, float("nan"))
            _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, 4, 4, 3, Wa, _stream()), "fwd")
            _lib.check(lib.ct_gconv_bwd_data(_ptr(x), _ptr(w), _ptr(gx), B, G, 4, 4, 3, Wa, _stream()), "bwd_data")
            outs[flag] = (y, gx)
    finally:
        lib.ct_debug_set_gconv(0)
    if B * G * W[0] * W[1] * W[2] <= (1 << 20):               # float64 on the CPU for the small cases
        xd = x.double().cpu().requires_grad_(True)
        yr = torch.nn.functional.conv3d(xd, w.double().cpu(), b.double().cpu(), padding=1, groups=G)
        gxr, = torch.autograd.grad(yr, xd, x.double().cpu())
        for flag in (4, 2):
            assert float((outs[flag][0].double().cpu() - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
            assert float((outs[flag][1].double().cpu() - gxr).abs().max()) <= 2e-5 * float(gxr.abs().max())
    for k in range(2):
        a, r = outs[4][k], outs[2][k]
        assert torch.isfinite(a).all()
        assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())
    # weight gradient: gconv_c4_wrw_mfma3_kernel (rows (dz, ci) x columns (dx, co), one MFMA per row tap; taken whenever its plan
    # fits) against the ring kernel's vector-ALU engine, twice (bitwise reproducible)
    gy = torch.randn_like(x)
    nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, 4, 4, 3, Wa)
    ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
    res = []
    try:
        for flag in (0, 0, 2):
            lib.ct_debug_set_gconv(flag)
            gw = torch.full((G * 4, 4, 3, 3, 3), float("nan"), device="cuda")
            gb = torch.full((G * 4,), float("nan"), device="cuda")
            _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), nws, B, G, 4, 4, 3, Wa, _stream()), "wrw")
            res.append((gw, gb))
    finally:
        lib.ct_debug_set_gconv(0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in range(2):
        assert torch.isfinite(res[0][k]).all()
        assert float((res[0][k] - res[2][k]).abs().max()) <= 2e-5 * float(res[2][k].abs().max())


@pytest.mark.parametrize("B,G,Ci,Co,W", [(8, 16, 64, 64, (8, 8, 8)), (8, 16, 32, 64, (8, 8, 8)), (3, 5, 48, 40, (4, 8, 8)), (1, 2, 64, 64, (4, 4, 4)),
                                         (8, 16, 64, 64, (16, 16)), (2, 3, 32, 64, (8, 8)), (5, 2, 64, 32, (4, 4)), (2, 16, 64, 64, (8, 8, 8))])
def test_small_volume_weight_gradient_on_the_matrix_cores(B, G, Ci, Co, W):
    """gconv_wrw_mfma_kernel (K = 4 x positions per MFMA, batch split over workgroups + fixed-order reduction when the channel
    blocks do not cover the chip) against the vector-ALU form and float64; bitwise reproducible."""
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.ops import _ptr, _stream
    lib = _lib.load()
    dim = len(W)
    torch.manual_seed(Ci + Co + B)
    x = torch.randn(B, G * Ci, *W, device="cuda")
    gy = torch.randn(B, G * Co, *W, device="cuda")
    Wa = _lib.int_array(W)
    nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, Ci, Co, dim, Wa)
    ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
    res = []
    try:
        for flag in (0, 0, 1):
            lib.ct_debug_set_gconv(flag)
            gw = torch.full((G * Co, Ci) + (3,) * dim, float("nan"), device="cuda")
            gb = torch.full((G * Co,), float("nan"), device="cuda")
            _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), nws, B, G, Ci, Co, dim, Wa, _stream()), "wrw")
            res.append((gw, gb))
    finally:
        lib.ct_debug_set_gconv(0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    fn = torch.nn.functional.conv3d if dim == 3 else torch.nn.functional.conv2d
    wd = torch.zeros((G * Co, Ci) + (3,) * dim, dtype=torch.float64, requires_grad=True)
    bd = torch.zeros(G * Co, dtype=torch.float64, requires_grad=True)
    yr = fn(x.double().cpu(), wd, bd, padding=1, groups=G)
    gwr, gbr = torch.autograd.grad(yr, (wd, bd), gy.double().cpu())
    for gw, gb in (res[0], res[2]):
        assert float((gw.double().cpu() - gwr).abs().max()) <= 2e-5 * float(gwr.abs().max())
        assert float((gb.double().cpu() - gbr).abs().max()) <= 2e-5 * float(gbr.abs().max())


def _rand_cfg_wide(rng):
    """wide groups (24 .. 80 channels per side) on small volumes: the K-split forward / backward-data with its LDS-DMA and
    element-wise bank staging, the matrix-core weight gradient with and without the batch split, ragged 16-channel blocks"""
    dim = int(rng.integers(2, 4))
    Wx = int(rng.choice([4, 8, 16]))
    H = int(rng.integers(1, 9))
    D = int(rng.integers(1, 9)) if dim == 3 else None
    pick = lambda: int(rng.choice([32, 48, 64, 80])) if rng.random() < 0.5 else int(rng.integers(24, 81))      # noqa: E731
    W = (D, H, Wx) if dim == 3 else (H, Wx)
    return (int(rng.integers(1, 10)), int(rng.integers(1, 5)), pick(), pick(), dim, W, bool(rng.random() < 0.5))


@pytest.mark.parametrize("seed", range(16))
def test_gconv_fuzz_wide_groups(seed, cfg_of=_rand_cfg_wide, base=5000):
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.layers.gconv import GroupedConvFn
    rng = np.random.default_rng(base + seed)
    B, G, Cin, Cout, dim, W, bias = cfg_of(rng)
    if not _lib.load().ct_gconv_supported(B, G, Cin, Cout, dim, _lib.int_array(W)):
        pytest.skip("no tile plan: the module takes the library convolution")
    torch.manual_seed(seed)
    x = torch.randn(B, G * Cin, *W)
    w = torch.randn(G * Cout, Cin, *([3] * dim)) / (Cin * 3 ** dim) ** 0.5
    b = torch.randn(G * Cout) if bias else None
    cot = torch.randn(B, G * Cout, *W)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    fn = torch.nn.functional.conv3d if dim == 3 else torch.nn.functional.conv2d
    yr = fn(xr, wr, br, stride=1, padding=1, groups=G)
    (yr * cot.double()).sum().backward()
    xc, wc = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bc = b.cuda().requires_grad_(True) if bias else None
    y = GroupedConvFn.apply(xc, wc, bc, G)
    (y * cot.cuda()).sum().backward()

    def close(a, r, name, tol):
        err = float((a.detach().cpu().double() - r.detach()).abs().max())
        assert err <= tol * max(1.0, float(r.abs().max())), (name, (B, G, Cin, Cout, dim, W, bias), err)

    close(y, yr, "y", 2e-5)
    close(xc.grad, xr.grad, "g_x", 2e-5)
    close(wc.grad, wr.grad, "g_w", 5e-5)
    if bias:
        close(bc.grad, br.grad, "g_bias", 5e-5)


def _rand_cfg_c4_3d(rng):
    Wx = int(rng.choice([16, 32, 48, 64, 128]))
    return (int(rng.integers(1, 4)), int(rng.integers(1, 4)), 4, 4, 3, (int(rng.integers(1, 12)), int(rng.integers(1, 40)), Wx), bool(rng.random() < 0.5))


@pytest.mark.parametrize("seed", range(8))
def test_gconv_fuzz_four_channel_3d_on_the_matrix_cores(seed):
    from cloud_transformers_amd import _lib
    lib = _lib.load()
    lib.ct_debug_set_gconv(4)                                  # the matrix-core kernel whatever the size
    try:
        test_gconv_fuzz_wide_groups(seed, _rand_cfg_c4_3d, 6000)
    finally:
        lib.ct_debug_set_gconv(0)

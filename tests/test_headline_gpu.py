"""Parity of the kernels the headline number is made of, at the headline shape.

BENCH shape (BASELINE.json north star): B8 N4096 H64 32x32, C in {16, 4}, reduce max and sum.  The whole
fwd+bwd chain runs on the HIP path at full size; (b,h) planes are independent, so sampled planes are compared
with the oracle run on those planes alone — z, out, g_z (Slice backward's grid cotangent), g_feat, g_keys.
The test also asserts WHICH kernel family each entry point launched (ct_debug_last_launch), so that a change
of the dispatch rules cannot silently move the headline shape onto kernels this file does not check, and
compares the hot-shape kernels with the generic ones on full tensors (ct_debug_set_flags).

Bars: grid of Splat(max) bit-exact; every float sum within 1e-4 of the tensor's max; the fixed-point
scatter-add within 1e-4 of EACH CHANNEL's own max (per-channel quantum)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

NAMES = ("z", "out", "g_z", "g_feat", "g_keys")


def _lib():
    from cloud_transformers_amd import _lib
    return _lib, _lib.load()


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def hip_chain(keys, feat, cot, W, H, dim, reduce, pad=None):
    from cloud_transformers_amd import ops
    _, lib = _lib()
    k = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    tags = {}
    z = ops.splat_keys(k, f, pad, W, H, dim, reduce)
    tags["splat_fwd"] = lib.ct_debug_last_launch().decode()
    z.retain_grad()
    o = ops.slice_keys(k, z, pad, W, H, dim)
    tags["slice_fwd"] = lib.ct_debug_last_launch().decode()
    o.backward(cot)
    return (z.detach(), o.detach(), z.grad, f.grad, k.grad), tags


def oracle_chain(keys, feat, cot, W, H, dim, reduce, pad=None):
    k = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    lc, idx = R.positions(k, W, H, dim)
    z = R.splat(lc, idx, f, pad, W, H, dim, reduce)
    z.retain_grad()
    o = R.slice_(lc, idx, z, pad, W, H, dim)
    o.backward(cot)
    return z.detach(), o.detach(), z.grad, f.grad, k.grad


@pytest.fixture
def flags():
    mod, lib = _lib()
    yield lambda v: lib.ct_debug_set_flags(v)
    lib.ct_debug_set_flags(0)


PLANES = ((0, 0), (1, 63), (2, 5), (3, 17), (4, 40), (5, 31), (6, 9), (7, 63))


@pytest.mark.parametrize("reduce", ["max", "sum"])
@pytest.mark.parametrize("C", [16, 4])
def test_headline_shape_fwd_bwd_against_oracle_planes(C, reduce, flags):
    from cloud_transformers_amd.step import SplatSliceStep
    mod, lib = _lib()
    torch.manual_seed(1234)
    B, N, H, W, dim = 8, 4096, 64, 32, 2
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    # the unit bench.py times
    step = SplatSliceStep(keys, feat, cot, W, H, dim, reduce)
    step.run()
    torch.cuda.synchronize()
    tags = step.launch_tags()
    assert tags["slice_fwd"] == "gather_ci", tags
    # (512 planes and >= 3 channel groups: the sorted-plane form, csrc/ct_raster_sorted.h — tests/test_sorted_gpu.py compares
    #  it with the scatter form; C4: the scatter form)
    assert tags["slice_bwd"] == ("slice_bwd_sorted" if C >= 12 else "slice_bwd_fused"), tags
    assert tags["splat_fwd"] == ("scatter_quad_max" if reduce == "max" else "scatter_add_sorted" if C >= 12 else "scatter_add_fused"), tags
    assert tags["splat_bwd"] == ("splat_max_bwd_hot" if reduce == "max" else "splat_sum_bwd_hot"), tags
    step.run()                                    # (launch_tags re-ran the passes: g_keys was accumulated twice)
    torch.cuda.synchronize()
    got = (step.z, step.out, step.g_z, step.g_feat, step.g_keys())
    # ... equals the module path (separate g_keys of Splat and Slice summed by autograd)
    via_ops, _ = hip_chain(keys, feat, cot, [W, W], H, dim, reduce)
    for name, a, b in zip(NAMES, got, via_ops):
        assert relerr(a, b) <= (0.0 if name in ("z", "out", "g_z", "g_feat") else 1e-6), name
    # sampled planes against the oracle
    for (b, h) in PLANES:
        ref = oracle_chain(keys[b:b + 1, h * 2:(h + 1) * 2].cpu(), feat[b:b + 1, h * C:(h + 1) * C].cpu(),
                           cot[b:b + 1, h * C:(h + 1) * C].cpu(), [W, W], 1, dim, reduce)
        sl = slice(h * C, (h + 1) * C)
        mine = (got[0][b:b + 1, sl], got[1][b:b + 1, sl], got[2][b:b + 1, sl], got[3][b:b + 1, sl],
                got[4][b:b + 1, h * 2:(h + 1) * 2])
        if reduce == "max":
            assert torch.equal(mine[0].cpu(), ref[0]), "z plane (%d,%d)" % (b, h)          # bit-exact
        for name, a, r in zip(NAMES, mine, ref):
            assert relerr(a, r) <= 1e-4, "%s plane (%d,%d): %.2e" % (name, b, h, relerr(a, r))
    # hot-shape kernels vs the generic kernels, full tensors
    flags(mod.DEBUG_NO_HOT)
    gen, gtags = hip_chain(keys, feat, cot, [W, W], H, dim, reduce)
    flags(0)
    assert gtags["slice_fwd"] == "gather_quad", gtags
    assert torch.equal(got[0], gen[0]) if reduce == "max" else relerr(got[0], gen[0]) <= 1e-6
    for name, a, b in zip(NAMES[1:], got[1:], gen[1:]):
        assert relerr(a, b) <= 1e-5, name


SMALL = [
    # B, H, C, N, W, pad, duplicated points
    (2, 3, 8, 1024, (32, 32), False, False),
    (1, 2, 16, 4096, (32, 32), False, False),     # two quads per thread
    (1, 2, 4, 8192, (32, 32), False, False),      # Splat bwd: g_keys through memory; Slice bwd not fused (N > 4096)
    (2, 2, 12, 516, (16, 24), True, False),       # non-square, padding mask, chunk of 12 channels
    (1, 2, 8, 256, (8, 8), False, True),          # exact ties: the claims pass of Splat(max) backward
    (1, 1, 20, 2048, (16, 16), True, False),
    (1, 2, 8, 2052, (32, 32), False, False),      # ragged tail of the second quad
    (1, 1, 32, 4096, (48, 40), False, False),     # several chunks, whole-CU LDS
    # 3D (trilinear) forms: the zoo's volume heads
    (2, 2, 16, 1024, (16, 16, 16), False, False),
    (1, 2, 32, 4096, (8, 8, 8), False, False),    # two quads per thread, 32 channels
    (1, 2, 8, 8192, (8, 8, 8), False, False),     # g_keys through memory
    (2, 2, 12, 516, (6, 8, 10), True, False),     # non-cubic, padding mask, ragged tail
    (1, 2, 8, 256, (4, 4, 4), False, True),       # exact ties in 3D
]


@pytest.mark.parametrize("reduce", ["max", "sum"])
@pytest.mark.parametrize("cfg", SMALL, ids=[str(c) for c in SMALL])
def test_hot_kernels_forced_on_small_shapes(cfg, reduce, flags):
    """The hot-shape kernels on shapes the oracle handles whole (they are normally reserved for >= 128 planes)."""
    mod, lib = _lib()
    B, H, C, N, W, use_pad, dup = cfg
    dim = len(W)
    g = torch.Generator().manual_seed(B * 131 + C * 7 + N)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    if dup:
        keys = keys[..., : N // 2].repeat(1, 1, 2)
        feat = feat[..., : N // 2].repeat(1, 1, 2)
    cot = torch.randn(B, H * C, N, generator=g)
    pad = (torch.rand(B, N, generator=g) > 0.2).float() if use_pad else None
    ref = oracle_chain(keys, feat, cot, list(W), H, dim, reduce, pad)
    flags(mod.DEBUG_FORCE_HOT)
    got, tags = hip_chain(keys.cuda(), feat.cuda(), cot.cuda(), list(W), H, dim, reduce, None if pad is None else pad.cuda())
    flags(0)
    assert tags["slice_fwd"] == ("gather_ci" if dim == 2 else "gather_ci3"), tags
    if reduce == "max":
        assert torch.equal(got[0].cpu(), ref[0])
    if dup and reduce == "max":
        # which of two identical points wins a cell is unspecified (torch's amax backward even splits it): the two
        # copies' gradients sum to the same total
        h = N // 2
        for name, a, r in zip(NAMES[:3], got[:3], ref[:3]):
            assert relerr(a, r) <= 1e-4, name
        for a, r in ((got[3].cpu(), ref[3]), (got[4].cpu(), ref[4])):
            assert relerr(a[..., :h] + a[..., h:], r[..., :h] + r[..., h:]) <= 1e-4
        gf = got[3].cpu()
        both = (gf[..., :h] != 0) & (gf[..., h:] != 0)
        # single winner per (cell, channel): a point's two copies may each win DIFFERENT cells, but the total routed to
        # a cell's channel equals the cotangent once — checked through the sums above; here: not every pair is split
        assert int(both.sum()) < int((gf[..., :h] != 0).sum())
    else:
        for name, a, r in zip(NAMES, got, ref):
            assert relerr(a, r) <= 1e-4, "%s: %.2e" % (name, relerr(a, r))


def _per_channel_err(got, ref, C):
    """max over channels of max|got - ref| / max|ref| of that channel; tensors [B, H*C, ...]"""
    g = got.detach().cpu().double().flatten(2)
    r = ref.detach().cpu().double().flatten(2)
    err = (g - r).abs().amax(dim=2)
    scale = r.abs().amax(dim=2).clamp_min(1e-300)
    return float((err / scale).max())


@pytest.mark.parametrize("family", ["hot", "generic", "explicit"])
def test_fixed_point_scatter_add_keeps_every_channels_precision(family, flags):
    """Channels of one head whose magnitudes differ by 10^8, on a clustered cloud (thousands of contributions to a
    cell): the scatter-add's error is judged PER CHANNEL against that channel's own max (1e-4), for Slice backward's
    grid cotangent and for Splat(sum) — the quantum of the fixed-point accumulation is per channel."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    g = torch.Generator().manual_seed(99)
    B, H, C, N, W, dim = 2, 2, 16, 4096, [32, 32], 2
    # three quarters of the cloud sit in one cell's neighbourhood
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    keys[..., : 3 * N // 4] = 0.31 + 0.01 * torch.randn(B, H * dim, 3 * N // 4, generator=g)
    scales = 10.0 ** torch.linspace(-4, 4, C)
    src = torch.randn(B, H * C, N, generator=g) * scales.repeat(H)[None, :, None]
    lc, idx = R.positions(keys, W, H, dim)
    ref = R.splat(lc.double(), idx, src.double(), None, W, H, dim, "sum") if False else None
    # float64 reference of the scatter-add (the fp32 oracle's own rounding is ~1e-7 per channel too)
    Bv, HC = src.shape[:2]
    G = W[0] * W[1]
    index = idx[:, :, None].reshape(B, H, 1, -1).expand(B, H, C, -1)
    pre = (src.double().reshape(B, H, C, N)[:, :, :, None] * lc.double()[:, :, None]).reshape(B, H, C, -1)
    ref = torch.zeros(B, H, C, G, dtype=torch.float64).scatter_add(3, index, pre).reshape(B, H * C, *W)

    flags({"hot": mod.DEBUG_FORCE_HOT, "generic": mod.DEBUG_NO_HOT, "explicit": mod.DEBUG_NO_HOT}[family])
    kc, sc = keys.cuda(), src.cuda()
    if family == "explicit":
        lcd, idxd = ops.positions(kc, W, H, dim)
        z = ops.splat_lc(lcd, idxd, sc, None, W, H, dim, "sum")
        grid = torch.zeros(B, H * C, *W, device="cuda", requires_grad=True)
        ops.slice_lc(lcd, idxd, grid, None, W, H, dim).backward(sc)
    else:
        z = ops.splat_keys(kc, sc, None, W, H, dim, "sum")
        grid = torch.zeros(B, H * C, *W, device="cuda", requires_grad=True)
        ops.slice_keys(kc, grid, None, W, H, dim).backward(sc)
        tag = lib.ct_debug_last_launch().decode()
        # (few planes: the channel chunks of a plane are dealt to several workgroups — "..._groups")
        # ("+folded": the groups' partial g_keys added inside the kernel, arrival tickets — tests/test_tickets_gpu.py)
        assert tag.replace("+folded", "") == ("slice_bwd_fused_groups" if family == "hot" else "slice_bwd_gw_stats_parts+scatter_quad_add"), tag
    flags(0)
    assert _per_channel_err(z, ref, C) <= 1e-4, _per_channel_err(z, ref, C)
    assert _per_channel_err(grid.grad, ref, C) <= 1e-4, _per_channel_err(grid.grad, ref, C)


def test_splat_bwd_accumulates_into_g_keys(flags):
    """ct_splat_bwd_ex(CT_BWD_ACCUMULATE_KEYS): g_keys += result in every form of the Splat(max) backward — the hot
    kernel and the whole-plane / quad / generic kernels add in their own store, chunk groups accumulate in the ordered
    sum of their partials (or with atomics when the workspace has no room for them)."""
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    HOT, NOHOT = mod.DEBUG_FORCE_HOT, mod.DEBUG_NO_HOT
    cases = ((2, 2, 8, 1024, 32, 2, HOT, "splat_max_bwd_hot"), (2, 2, 8, 1024, 32, 2, 0, "splat_max_bwd_quad"),
             (1, 2, 5, 333, 16, 2, 0, "splat_max_bwd_generic"), (1, 2, 32, 2048, 64, 2, NOHOT, "splat_max_bwd_quad"),
             (16, 16, 16, 512, 32, 2, NOHOT, "splat_max_bwd_whole_head"), (2, 2, 8, 1024, 16, 3, 0, "splat_max_bwd_generic"),
             (1, 2, 16, 512, 16, 3, NOHOT, "splat_max_bwd_generic"))
    for (B, H, C, N, W, dim, dbg, want_tag) in cases:
        g = torch.Generator().manual_seed(N + C)
        Ws = [W] * dim
        keys = torch.tanh(torch.randn(B, H * dim, N, generator=g)).cuda()
        feat = torch.randn(B, H * C, N, generator=g).cuda()
        gz = torch.randn(B, H * C, *Ws, generator=g).cuda()
        Wa = mod.int_array(Ws)
        z = torch.empty(B, H * C, *Ws, device="cuda")
        mod.check(lib.ct_splat_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(z), B, H, C, N, dim, Wa, 0, _stream()), "fwd")
        base = torch.randn(B, H * dim, N, generator=g).cuda()
        gk_plain = torch.empty_like(keys)
        gf_a, gf_b = torch.empty_like(feat), torch.empty_like(feat)
        n0 = lib.ct_splat_bwd_workspace_bytes(B, H, C, N, dim, Wa, 0)
        ws0 = torch.empty(max(n0, 1), device="cuda", dtype=torch.uint8)
        mod.check(lib.ct_splat_bwd(_ptr(keys), _ptr(feat), None, 0, _ptr(z), _ptr(gz), _ptr(gf_a), _ptr(gk_plain),
                                   _ptr(ws0), n0, B, H, C, N, dim, Wa, 0, _stream()), "bwd")
        n1 = lib.ct_splat_bwd_ex_workspace_bytes(B, H, C, N, dim, Wa, 0, mod.BWD_ACCUMULATE_KEYS)
        assert n1 == n0 + keys.numel() * 4
        for nws in (n1, keys.numel() * 4):         # full workspace; only the mandatory tail (no room for partial sums)
            ws1 = torch.empty(nws, device="cuda", dtype=torch.uint8)
            gk_acc = base.clone()
            flags(dbg)
            mod.check(lib.ct_splat_bwd_ex(_ptr(keys), _ptr(feat), None, 0, _ptr(z), _ptr(gz), _ptr(gf_b), _ptr(gk_acc),
                                          _ptr(ws1), nws, B, H, C, N, dim, Wa, 0, mod.BWD_ACCUMULATE_KEYS, _stream()), "bwd_ex")
            tag = lib.ct_debug_last_launch().decode()
            flags(0)
            if nws == n1:
                assert tag.startswith(want_tag), (tag, want_tag)
            assert "add_inplace" not in tag, tag
            assert relerr(gk_acc, base + gk_plain) <= 2e-6, (B, H, C, N, W, dim, tag)
            assert relerr(gf_b, gf_a) <= 1e-6
        # too small a workspace is refused
        assert lib.ct_splat_bwd_ex(_ptr(keys), _ptr(feat), None, 0, _ptr(z), _ptr(gz), _ptr(gf_b), _ptr(gk_acc),
                                   _ptr(ws1), 16, B, H, C, N, dim, Wa, 0, mod.BWD_ACCUMULATE_KEYS, _stream()) == -3


@pytest.mark.parametrize("W", [(32, 32), (8, 8, 8)], ids=["2d", "3d"])
def test_fused_slice_backward_with_non_finite_channels(W, flags):
    """The fused Slice backward accumulates channel PAIRS in 64-bit fixed-point words; a channel that holds inf / NaN (or
    would overflow the bound) takes IEEE float atomics — and so does its pair partner, which must stay exact to float
    accuracy, as must every other channel."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    dim = len(W)
    B, H, C, N = 1, 2, 8, 1024
    g = torch.Generator().manual_seed(5 + dim)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    grid = torch.randn(B, H * C, *W, generator=g)
    cot = torch.randn(B, H * C, N, generator=g)
    cot[0, 1, 7] = float("inf")                 # head 0, channel 1: partner of channel 0
    cot[0, 4, 100] = float("nan")               # head 0, channel 4: partner of channel 5
    cot[0, C + 6] *= 1e30                        # head 1, channel 6: finite, but a sum of it could overflow
    lc, idx = R.positions(keys, list(W), H, dim)
    ref = torch.zeros(B, H * C, *W)
    gr = grid.clone().requires_grad_(True)
    R.slice_(lc, idx, gr, None, list(W), H, dim).backward(cot)
    ref = gr.grad
    flags(mod.DEBUG_FORCE_HOT)
    gk = grid.cuda().requires_grad_(True)
    ops.slice_keys(keys.cuda(), gk, None, list(W), H, dim).backward(cot.cuda())
    tag = lib.ct_debug_last_launch().decode()
    flags(0)
    assert tag.startswith("slice_bwd_fused"), tag
    got = gk.grad.cpu()
    for ch in range(H * C):
        a, r = got[0, ch].double(), ref[0, ch].double()
        if ch in (1, 4):
            # the same cells are inf / NaN, every other cell of the channel agrees
            assert torch.equal(torch.isfinite(a), torch.isfinite(r)), ch
            fin = torch.isfinite(r)
            assert float((a[fin] - r[fin]).abs().max()) <= 1e-4 * float(r[fin].abs().max()), ch
        else:
            assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()), ch


ZOO_SHAPES = [
    # C, W, dim, B, N  — the six zoo head shapes at the S3DIS batch (model_zoo/s3dis/segmenter.py:28-45) and two decoder
    # shapes of the completion model (B2 N16384, model_zoo/completion/inpainter.py:135-155)
    (4, 128, 2, 8, 4096), (4, 32, 3, 8, 4096), (16, 64, 2, 8, 4096), (16, 16, 3, 8, 4096), (16, 16, 2, 8, 4096),
    (32, 8, 3, 8, 4096), (16, 64, 2, 2, 16384), (32, 8, 3, 2, 16384), (16, 16, 2, 2, 16384), (16, 16, 3, 2, 16384),
    # round 5: the shapes that used to run only in tools/zoo_sweep.py — the ScanObjectNN batch
    # (BASELINE configs[1]: B8 N2048, model_zoo/scanobject/classifier.py:46-92) on all six head shapes, the What3D batch
    # (configs[4]: B4 N8192) on the decoder's widest head, and the two four-channel decoder heads at B2 N16384
    (4, 128, 2, 8, 2048), (4, 32, 3, 8, 2048), (16, 64, 2, 8, 2048), (16, 16, 3, 8, 2048), (16, 16, 2, 8, 2048),
    (32, 8, 3, 8, 2048), (16, 64, 2, 4, 8192), (4, 128, 2, 2, 16384), (4, 32, 3, 2, 16384),
]


@pytest.mark.parametrize("cfg", ZOO_SHAPES, ids=[str(c) for c in ZOO_SHAPES])
def test_zoo_head_shapes_at_full_size_against_oracle_planes(cfg, flags):
    """The kernels the zoo's heads actually run (H = 16: chunk groups, partial g_keys sums, 3D hot kernels, split-N
    statistics kernels) at full size in the default dispatch: the whole fwd+bwd step against the oracle on sampled (b, h)
    planes, and the other kernel family on full tensors."""
    from cloud_transformers_amd.step import SplatSliceStep
    mod, lib = _lib()
    C, Wn, dim, B, N = cfg
    H, W = 16, [Wn] * dim
    torch.manual_seed(77 + C + Wn)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    step = SplatSliceStep(keys, feat, cot, Wn, H, dim, "max")
    step.run()
    torch.cuda.synchronize()
    got = (step.z, step.out, step.g_z, step.g_feat, step.g_keys().clone())
    step.slice_bwd()                               # (Splat's key cotangent may have been added in place: Slice's alone, again)
    torch.cuda.synchronize()
    gk_slice_all = step.g_keys_buf.clone()
    for (b, h) in ((0, 0), (B - 1, H - 1), (B // 2, 5)):
        ref = oracle_chain(keys[b:b + 1, h * dim:(h + 1) * dim].cpu(), feat[b:b + 1, h * C:(h + 1) * C].cpu(),
                           cot[b:b + 1, h * C:(h + 1) * C].cpu(), W, 1, dim, "max")
        sl = slice(h * C, (h + 1) * C)
        mine = (got[0][b:b + 1, sl], got[1][b:b + 1, sl], got[2][b:b + 1, sl], got[3][b:b + 1, sl],
                got[4][b:b + 1, h * dim:(h + 1) * dim])
        assert torch.equal(mine[0].cpu(), ref[0]), "z plane (%d,%d)" % (b, h)
        # A point that lies EXACTLY on a cell boundary has corner weights of exactly 0: its product into the far cell is
        # +-0.0, and where that cell stays at the zero floor the three implementations disagree on purpose — torch_scatter
        # routes the cell's cotangent to the zero-valued candidate, torch's amax (the oracle) splits it with the floor, the
        # kernels here route nothing (HISTORY.md §2, INTEGRATION.md "Known deviations").  Only that point's key cotangent can differ (0 * x carries no
        # feature gradient): such points (a few per million) are left out of the g_keys comparison.
        lc_ref, _ = R.positions(keys[b:b + 1, h * dim:(h + 1) * dim].cpu(), W, 1, dim)
        on_edge = (lc_ref[0, 0] == 0).any(dim=0)                       # (N,)
        assert int(on_edge.sum()) <= max(2, N // 2000), int(on_edge.sum())
        for name, a, r in zip(NAMES, mine, ref):
            if name == "g_keys":
                a, r = a.cpu()[..., ~on_edge], r[..., ~on_edge]
            assert relerr(a, r) <= 1e-4, "%s plane (%d,%d): %.2e" % (name, b, h, relerr(a, r))
        # The mask above is for the SPLAT(max) term alone: Slice's key cotangent — what ct_slice_bwd* left in
        # g_keys_buf before Splat's was added into g_keys_out — has no such deviation and is held to the oracle on EVERY point
        k2 = keys[b:b + 1, h * dim:(h + 1) * dim].cpu().clone().requires_grad_(True)
        lc2, idx2 = R.positions(k2, W, 1, dim)
        R.slice_(lc2, idx2, ref[0], None, W, 1, dim).backward(cot[b:b + 1, h * C:(h + 1) * C].cpu())
        gk_slice = gk_slice_all[b:b + 1, h * dim:(h + 1) * dim]
        assert relerr(gk_slice, k2.grad) <= 1e-4, "Slice's g_keys, all points, plane (%d,%d): %.2e" % (b, h, relerr(gk_slice, k2.grad))
    # the other kernel family on the full tensors (module path: separate g_keys of Splat and Slice summed by autograd)
    flags(mod.DEBUG_NO_HOT)
    gen, _ = hip_chain(keys, feat, cot, W, H, dim, "max")
    flags(0)
    assert torch.equal(got[0], gen[0])
    for name, a, b_ in zip(NAMES[1:], got[1:], gen[1:]):
        assert relerr(a, b_) <= 1e-5, name


@pytest.mark.parametrize("dim,Wn,C,N", [(2, 16, 8, 6156), (3, 8, 8, 6156), (2, 32, 16, 8192)])
def test_fused_slice_backward_over_point_segments(dim, Wn, C, N, flags):
    """Clouds longer than a workgroup's 4096 points: the fused Slice backward cuts every plane into equal segments (6156 =
    3 x 2052, 8192 = 2 x 4096), one workgroup each with its own partial g_grid tile and its own fixed-point quanta, and adds
    the partial tiles in a fixed order — same results as the generic kernels, bitwise reproducible, padding mask honoured."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    B, H, W = 2, 16, [Wn] * dim
    g = torch.Generator().manual_seed(5 + N + dim)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g)).cuda()
    grid = torch.randn(B, H * C, *W, generator=g).cuda()
    cot = torch.randn(B, H * C, N, generator=g).cuda()
    pad = (torch.rand(B, N, generator=g) > 0.2).float().cuda()
    res = {}
    for fam, fl in (("hot", 0), ("hot2", 0), ("generic", mod.DEBUG_NO_HOT)):
        flags(fl)
        k = keys.clone().requires_grad_(True)
        gr = grid.clone().requires_grad_(True)
        ops.slice_keys(k, gr, pad, W, H, dim).backward(cot)
        res[fam] = (gr.grad, k.grad, lib.ct_debug_last_launch().decode())
    flags(0)
    assert "segments" in res["hot"][2], res["hot"][2]
    assert "segments" not in res["generic"][2]
    assert torch.equal(res["hot"][0], res["hot2"][0])            # bitwise reproducible
    assert _per_channel_err(res["hot"][0], res["generic"][0], C) <= 1e-4
    assert relerr(res["hot"][1], res["generic"][1]) <= 1e-5

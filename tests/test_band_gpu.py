"""Banded backward kernels of four-channel heads (csrc/ct_raster_band.h: Slice backward fused, Splat(max) backward): the
dispatch gives them the 2D grids whose single-channel tile exceeds a CU's LDS (256^2 ...: otherwise global atomics); on the
zoo's 128^2 C4 / 32^3 C4 heads they were measured slower than the single-channel kernels and run only when forced.
Against the oracle on shapes it handles whole (forced onto small grids: several bands, a short last band, padding masks,
ragged scan rounds, exact ties inside a band and across the row two bands share), and against the other kernel families
at full sizes.

Bars: g_grid within 1e-4 of each channel's max (fixed-point scatter-add, per-channel quantum), g_feat / g_keys within
1e-4 (oracle) resp. 1e-5 (other kernel family) of the tensor's max; ties: one winner per (cell, channel)."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _libs():
    from cloud_transformers_amd import _lib
    return _lib, _lib.load()


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def per_channel_err(a, b, C):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    a, b = a.reshape(a.shape[0], -1, C, a[0, 0].numel()), b.reshape(b.shape[0], -1, C, b[0, 0].numel())
    return float(((a - b).abs().amax(3) / b.abs().amax(3).clamp_min(1e-30)).max())


def run_bwd(keys, feat, cot_grid, cot_pts, add, pad, W, H, dim, flags):
    """z, g_y (Slice bwd of cot_pts through grid = z), g_keys_slice, g_feat, g_keys (= add + Splat bwd of cot_grid); the tags."""
    from cloud_transformers_amd.ops import _pad_args, _ptr, _stream
    mod, lib = _libs()
    B, HC, N = feat.shape
    C = HC // H
    Wa = mod.int_array(W)
    padt, pad_code = _pad_args(pad, B, N)
    z = torch.empty(B, HC, *W, device="cuda")
    mod.check(lib.ct_splat_fwd(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(z), B, H, C, N, dim, Wa, 0, _stream()), "fwd")
    lib.ct_debug_set_flags(flags)
    try:
        nws = max(lib.ct_slice_bwd_workspace_bytes(B, H, C, N, dim, Wa), lib.ct_splat_bwd_ex_workspace_bytes(B, H, C, N, dim, Wa, 0, 1), 16)
        ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
        g_y = torch.full_like(z, float("nan"))
        gk_s = torch.full_like(keys, float("nan"))
        mod.check(lib.ct_slice_bwd_tk(_ptr(keys), _ptr(z), _ptr(padt), pad_code, _ptr(cot_pts), _ptr(g_y), _ptr(gk_s), _ptr(ws), nws,
                                      None, B, H, C, N, dim, Wa, _stream()), "slice_bwd")
        t1 = lib.ct_debug_last_launch().decode()
        g_feat = torch.full_like(feat, float("nan"))
        g_keys = torch.full_like(keys, float("nan"))
        mod.check(lib.ct_splat_bwd_tk(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(z), _ptr(cot_grid), _ptr(g_feat), _ptr(add),
                                      _ptr(g_keys), _ptr(ws), nws, None, B, H, C, N, dim, Wa, 0, _stream()), "splat_bwd")
        t2 = lib.ct_debug_last_launch().decode()
        torch.cuda.synchronize()
    finally:
        lib.ct_debug_set_flags(0)
    return (z, g_y, gk_s, g_feat, g_keys), (t1, t2)


def oracle_bwd(keys, feat, cot_grid, cot_pts, add, pad, W, H, dim):
    k1 = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    lc, idx = R.positions(k1, W, H, dim)
    z = R.splat(lc, idx, f, pad, W, H, dim, "max")
    (z * cot_grid).sum().backward()
    k2 = keys.clone().requires_grad_(True)
    zz = z.detach().clone().requires_grad_(True)
    lc2, idx2 = R.positions(k2, W, H, dim)
    R.slice_(lc2, idx2, zz, pad, W, H, dim).backward(cot_pts)
    return z.detach(), zz.grad, k2.grad, f.grad, add + k1.grad


SMALL = [
    # B, H, N, W, pad
    (2, 3, 1024, (16, 24), False),
    (1, 2, 2052, (20, 20), True),          # short last band, ragged scan round, padding mask
    (1, 1, 4096, (64, 8), False),          # many bands of a narrow grid
    (2, 2, 516, (12, 16), True),
    (1, 2, 4096, (8, 8, 8), False),        # 3D: slabs
    (2, 1, 1028, (6, 4, 8), True),
    (1, 1, 2048, (16, 4, 4), False),
]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: "B%dH%dN%d_%s_%s" % (c[0], c[1], c[2], "x".join(map(str, c[3])), "pad" if c[4] else "nopad"))
def test_banded_kernels_on_small_grids_against_the_oracle(case):
    mod, lib = _libs()
    B, H, N, W, with_pad = case
    W, dim, C = list(W), len(W), 4
    g = torch.Generator().manual_seed(sum(W) * 7 + N)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g) * 0.8)
    keys[0, 0, :8] = torch.tensor([1.0, -1.0, 0.99999994, -0.99999994, 0.0, 2.0, -3.0, 0.5])     # edges, clamped keys
    feat = torch.randn(B, H * C, N, generator=g)
    cot_grid = torch.randn(B, H * C, *W, generator=g)
    cot_pts = torch.randn(B, H * C, N, generator=g)
    cot_pts[:, 1::4] *= 1e-6                                   # a quiet channel keeps its own quantum
    add = torch.randn(B, H * dim, N, generator=g)
    pad = (torch.rand(B, N, generator=g) > 0.2).float() if with_pad else None
    dev = lambda t: None if t is None else t.cuda()
    got, tags = run_bwd(dev(keys), dev(feat), dev(cot_grid), dev(cot_pts), dev(add), dev(pad), W, H, dim, mod.DEBUG_FORCE_BAND)
    assert tags[0].startswith("band_slice_bwd") and tags[1].startswith("band_splat_bwd"), tags
    ref = oracle_bwd(keys, feat, cot_grid, cot_pts, add, pad, W, H, dim)
    assert torch.equal(got[0].cpu(), ref[0])
    assert per_channel_err(got[1], ref[1], C) <= 1e-4, per_channel_err(got[1], ref[1], C)
    for name, a, r in zip(("g_keys_slice", "g_feat", "g_keys"), got[2:], ref[2:]):
        assert relerr(a, r) <= 1e-4, (name, relerr(a, r))


@pytest.mark.parametrize("dim", [2, 3])
def test_banded_splat_backward_with_exact_ties(dim):
    """Duplicated points: every copy's product is bit-equal to z.  One winner per (cell, channel) — also for the cells of a
    row two bands share (both bands must pick the SAME winner: the lowest point index) — so the copies' gradients add up to
    the gradient of the de-duplicated cloud, and nothing is awarded twice."""
    mod, lib = _libs()
    B, H, C = 1, 2, 4
    W = [32, 16] if dim == 2 else [16, 4, 8]
    half = 1024
    g = torch.Generator().manual_seed(5 + dim)
    keys = torch.tanh(torch.randn(B, H * dim, half, generator=g))
    feat = torch.randn(B, H * C, half, generator=g)
    keys2, feat2 = keys.repeat(1, 1, 2), feat.repeat(1, 1, 2)
    N = 2 * half
    cot_grid = torch.randn(B, H * C, *W, generator=g)
    cot_pts = torch.randn(B, H * C, N, generator=g)
    add = torch.randn(B, H * dim, N, generator=g)
    got, tags = run_bwd(keys2.cuda(), feat2.cuda(), cot_grid.cuda(), cot_pts.cuda(), add.cuda(), None, W, H, dim, mod.DEBUG_FORCE_BAND)
    assert tags[1].startswith("band_splat_bwd"), tags
    k1 = keys.clone().requires_grad_(True)
    f1 = feat.clone().requires_grad_(True)
    lc, idx = R.positions(k1, W, H, dim)
    z1 = R.splat(lc, idx, f1, None, W, H, dim, "max")
    assert torch.equal(got[0].cpu(), z1.detach())
    (z1 * cot_grid).sum().backward()
    gf, gk = got[3].cpu(), got[4].cpu() - add
    assert not torch.isnan(gf).any() and not torch.isnan(gk).any()
    assert relerr(gf[..., :half] + gf[..., half:], f1.grad) <= 1e-5
    assert relerr(gk[..., :half] + gk[..., half:], k1.grad) <= 1e-4
    # the lowest point index wins every tie: the second copies receive nothing at all
    assert float(gf[..., half:].abs().max()) == 0.0


# the zoo's C4 heads (forced: their single-channel tiles fit LDS and the single-channel kernels are faster there, see
# band_plan) and two grids beyond a CU's LDS, which the banded kernels take by default instead of global atomics
ZOO = [(8, 4096, 16, 4, (128, 128), True), (2, 16384, 16, 4, (128, 128), True), (8, 4096, 16, 4, (32, 32, 32), True),
       (2, 16384, 16, 4, (32, 32, 32), True), (2, 8192, 4, 4, (256, 256), False), (1, 8192, 4, 4, (512, 256), False)]


@pytest.mark.parametrize("shape", ZOO, ids=lambda s: "B%dN%dH%dC%d_%s" % (s[0], s[1], s[2], s[3], "x".join(map(str, s[4]))))
def test_banded_kernels_at_full_sizes_against_the_other_kernel_families(shape):
    mod, lib = _libs()
    B, N, H, C, W, forced = shape
    W, dim = list(W), len(W)
    torch.manual_seed(7)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot_grid = torch.randn(B, H * C, *W, device="cuda")
    cot_pts = torch.randn(B, H * C, N, device="cuda")
    add = torch.randn(B, H * dim, N, device="cuda")
    fl = mod.DEBUG_FORCE_BAND if forced else 0
    got, tags = run_bwd(keys, feat, cot_grid, cot_pts, add, None, W, H, dim, fl)
    assert tags[0].startswith("band_slice_bwd") and tags[1].startswith("band_splat_bwd"), tags
    ref, rtags = run_bwd(keys, feat, cot_grid, cot_pts, add, None, W, H, dim, mod.DEBUG_NO_BAND)
    assert "band" not in rtags[0] and "band" not in rtags[1], rtags
    assert torch.equal(got[0], ref[0])
    assert per_channel_err(got[1], ref[1], C) <= 1e-4
    for name, a, r in zip(("g_keys_slice", "g_feat", "g_keys"), got[2:], ref[2:]):
        assert relerr(a, r) <= 1e-5, (name, relerr(a, r))
    # run to run: bitwise (fixed-point sums, every point and every cell owned by one workgroup)
    again, _ = run_bwd(keys, feat, cot_grid, cot_pts, add, None, W, H, dim, fl)
    for a, r in zip(got, again):
        assert torch.equal(a, r)

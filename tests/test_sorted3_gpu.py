"""Slice backward on sorted point segments, 3D grids (csrc/ct_raster_sorted3d.h).

Reference semantics: layers/cloud_transform.py:190-227 (Slice.forward: gather of the 2^3 corners, layers/utils.py:100-155, and the
weighted sum) — its backward is the scatter-add of the corner products into the grid and, through the weights, the key
cotangent (layers/cloud_transform.py:91-94).  Bars: g_grid within 1e-4 of EACH CHANNEL's own max (item sums rounded once to a
per-channel fixed-point quantum), g_keys within 1e-4; the sort is a pure function of the keys, so results are bitwise reproducible."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _lib():
    from cloud_transformers_amd import _lib
    return _lib, _lib.load()


@pytest.fixture
def flags():
    mod, lib = _lib()
    yield lambda v: lib.ct_debug_set_flags(v)
    lib.ct_debug_set_flags(0)


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def per_channel_err(a, b, HC):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    B = a.shape[0]
    a, b = a.reshape(B, HC, -1), b.reshape(B, HC, -1)
    return float(((a - b).abs().amax(dim=2) / b.abs().amax(dim=2).clamp_min(1e-30)).max())


def oracle(keys, z, cot, pad, W, H):
    dim = len(W)
    k = keys.clone().requires_grad_(True)
    zz = z.clone().requires_grad_(True)
    lc, idx = R.positions(k, list(W), H, dim)
    R.slice_(lc, idx, zz, pad, list(W), H, dim).backward(cot)
    return zz.grad, k.grad


def run(keys, z, cot, pad, W, H, C, tickets, fl):
    """-> (g_grid, g_keys, launch tag) through the C ABI, with the workspace the library asks for under these flags"""
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    B, _, N = keys.shape
    dim = len(W)
    Wa = mod.int_array(list(W))
    lib.ct_debug_set_flags(fl)
    try:
        nws = lib.ct_slice_bwd_workspace_bytes(B, H, C, N, dim, Wa)
        ws = torch.empty(max(nws, 16), device="cuda", dtype=torch.uint8)
        tk = torch.zeros(mod.TICKETS_BYTES // 4, device="cuda", dtype=torch.int32) if tickets else None
        g_z, g_k = torch.full_like(z, float("nan")), torch.full_like(keys, float("nan"))
        mod.check(lib.ct_slice_bwd_ps(_ptr(keys), _ptr(z), _ptr(pad), mod.PAD_F32 if pad is not None else 0, _ptr(cot), _ptr(g_z),
                                      _ptr(g_k), _ptr(ws), nws, _ptr(tk), None, B, H, C, N, dim, Wa, _stream()), "ct_slice_bwd_ps")
        tag = lib.ct_debug_last_launch().decode()
        torch.cuda.synchronize()
        if tickets:
            assert int(tk.abs().sum()) == 0, "the tickets were not handed back as zeros"
    finally:
        lib.ct_debug_set_flags(0)
    return g_z, g_k, tag


CASES = [
    # B, H, C, N, W, pad, duplicated points
    (2, 3, 8, 1024, (8, 8, 8), False, False),
    (1, 2, 16, 2048, (8, 8, 8), False, False),
    (1, 2, 8, 4096, (8, 8, 8), False, False),          # two point segments per plane: partial g_grid tiles
    (2, 2, 12, 516, (4, 6, 8), True, False),           # non-cubic, padding mask, ragged last quad of threads
    (1, 2, 8, 256, (4, 4, 4), False, True),
    (1, 1, 20, 2048, (8, 8, 8), True, False),
    (1, 2, 8, 2048, (8, 8, 8), False, True),           # heavy duplicates: cells with many items
    (1, 1, 8, 2048, (2, 2, 4), False, False),          # two base cells: the items of one cell fill whole waves
    (1, 2, 32, 6144, (8, 8, 8), False, False),         # three segments, eight groups
]


@pytest.mark.parametrize("cfg", CASES, ids=str)
@pytest.mark.parametrize("tickets", [False, True], ids=["sum_parts", "tickets"])
def test_sorted3_slice_backward_against_the_oracle(cfg, tickets):
    mod, lib = _lib()
    B, H, C, N, W, pad, dup = cfg
    torch.manual_seed(7)
    keys = torch.tanh(torch.randn(B, H * 3, N) * (0.3 if dup else 1.0))
    if dup:
        keys[:, :, N // 2:] = keys[:, :, :N // 2]
    keys[0, 0, :8] = torch.tensor([-1.0, 1.0, -0.99999994, 0.99999994, 0.0, 0.5, -2.0, 3.0])     # clamp edges (cotangent masked outside)
    z = torch.randn(B, H * C, *W)
    cot = torch.randn(B, H * C, N)
    p = (torch.rand(B, N) > 0.2).float() if pad else None
    gz_ref, gk_ref = oracle(keys, z, cot, p, W, H)
    kd, zd, cd = keys.cuda(), z.cuda(), cot.cuda()
    pd = p.cuda() if pad else None
    outs = []
    for _ in range(2):
        g_z, g_k, tag = run(kd, zd, cd, pd, W, H, C, tickets, mod.DEBUG_FORCE_SORTED | mod.DEBUG_FORCE_HOT)
        assert tag.startswith("slice_bwd_sorted3"), tag
        assert tag.endswith("+folded") == tickets or not tickets, tag
        outs.append((g_z, g_k))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "not bitwise reproducible"
    assert per_channel_err(outs[0][0], gz_ref, H * C) <= 1e-4
    assert relerr(outs[0][1], gk_ref) <= 1e-4


def test_sorted3_tickets_and_sum_parts_agree_bit_for_bit_and_with_the_scatter_form():
    """The zoo's 8^3 C32 head at B8 H16 N4096 (the default takes the sorted form): folds by tickets == folds by sum_parts launches
    bit for bit (the partials are added in ascending order either way); against the scatter form (slice_bwd_fused3) to 1e-5."""
    mod, lib = _lib()
    B, H, C, N, W = 8, 16, 32, 4096, (8, 8, 8)
    torch.manual_seed(3)
    keys = torch.tanh(torch.randn(B, H * 3, N, device="cuda"))
    z = torch.randn(B, H * C, *W, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    a = run(keys, z, cot, None, W, H, C, True, 0)
    b = run(keys, z, cot, None, W, H, C, False, 0)
    c = run(keys, z, cot, None, W, H, C, True, mod.DEBUG_NO_SORTED)
    assert a[2] == "slice_bwd_sorted3_segments+folded" and b[2] == "slice_bwd_sorted3_segments", (a[2], b[2])
    assert c[2].startswith("slice_bwd_fused3"), c[2]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert per_channel_err(a[0], c[0], H * C) <= 1e-5
    assert relerr(a[1], c[1]) <= 1e-5


def test_sorted3_slice_backward_with_non_finite_channels():
    """A channel that holds inf / NaN (or would overflow the fixed-point bound) takes IEEE float atomics; every other channel keeps
    its accuracy."""
    mod, lib = _lib()
    W, B, H, C, N = (8, 8, 8), 1, 2, 8, 1024
    g = torch.Generator().manual_seed(7)
    keys = torch.tanh(torch.randn(B, H * 3, N, generator=g))
    grid = torch.randn(B, H * C, *W, generator=g)
    cot = torch.randn(B, H * C, N, generator=g)
    cot[0, 1, 7] = float("inf")
    cot[0, 4, 100] = float("nan")
    cot[0, C + 6] *= 1e30
    ref, _ = oracle(keys, grid, cot, None, W, H)
    got, _, tag = run(keys.cuda(), grid.cuda(), cot.cuda(), None, W, H, C, False, mod.DEBUG_FORCE_SORTED | mod.DEBUG_FORCE_HOT)
    assert tag.startswith("slice_bwd_sorted3"), tag
    got = got.cpu()
    for ch in range(H * C):
        a, r = got[0, ch].double(), ref[0, ch].double()
        if ch in (1, 4):
            assert torch.equal(torch.isfinite(a), torch.isfinite(r)), ch
            fin = torch.isfinite(r)
            assert float((a[fin] - r[fin]).abs().max()) <= 1e-4 * float(r[fin].abs().max()), ch
        else:
            assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()), ch


@pytest.mark.parametrize("cfg", [c for c in CASES if c[3] <= 2048], ids=str)
def test_sorted3_splat_sum_forward_against_the_oracle(cfg, flags):
    """Splat(reduce=sum) forward on a small 3D grid = the scatter-add side of the sorted kernel alone (no conv tile, no key
    cotangent), one sorted segment per plane: layers/cloud_transform.py:164-173 with scatter_add_."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    B, H, C, N, W, pad, dup = cfg
    torch.manual_seed(9)
    keys = torch.tanh(torch.randn(B, H * 3, N) * (0.3 if dup else 1.0))
    if dup:
        keys[:, :, N // 2:] = keys[:, :, :N // 2]
    feat = torch.randn(B, H * C, N)
    p = (torch.rand(B, N) > 0.2).float() if pad else None
    lc, idx = R.positions(keys, list(W), H, 3)
    ref = R.splat(lc, idx, feat, p, list(W), H, 3, "sum")
    outs = []
    for _ in range(2):
        flags(mod.DEBUG_FORCE_SORTED | mod.DEBUG_FORCE_HOT)
        z = ops.splat_keys(keys.cuda(), feat.cuda(), p.cuda() if pad else None, list(W), H, 3, "sum")
        tag = lib.ct_debug_last_launch().decode()
        flags(0)
        assert tag == "scatter_add_sorted3", tag
        outs.append(z)
    assert torch.equal(outs[0], outs[1]), "not bitwise reproducible"
    assert per_channel_err(outs[0], ref, H * C) <= 1e-4


CASES_2D = [
    # B, H, C, N, W, pad, duplicated points — the same kernel with DIM = 2 (one face) on small 2D grids
    (2, 3, 8, 1024, (16, 16), False, False),
    (1, 2, 16, 4096, (16, 16), False, False),          # two point segments per plane
    (2, 2, 12, 516, (16, 24), True, False),            # non-square, padding mask, ragged last quad of threads
    (1, 2, 8, 2048, (8, 8), False, True),              # heavy duplicates
    (1, 1, 8, 6144, (16, 32), False, False),           # three segments on 512 cells
    (1, 2, 8, 1024, (4, 4), False, False),
]


@pytest.mark.parametrize("cfg", CASES_2D, ids=str)
@pytest.mark.parametrize("tickets", [False, True], ids=["sum_parts", "tickets"])
def test_sorted_segments_2d_slice_backward_against_the_oracle(cfg, tickets):
    mod, lib = _lib()
    B, H, C, N, W, pad, dup = cfg
    torch.manual_seed(11)
    keys = torch.tanh(torch.randn(B, H * 2, N) * (0.3 if dup else 1.0))
    if dup:
        keys[:, :, N // 2:] = keys[:, :, :N // 2]
    keys[0, 0, :8] = torch.tensor([-1.0, 1.0, -0.99999994, 0.99999994, 0.0, 0.5, -2.0, 3.0])
    z = torch.randn(B, H * C, *W)
    cot = torch.randn(B, H * C, N)
    p = (torch.rand(B, N) > 0.2).float() if pad else None
    gz_ref, gk_ref = oracle(keys, z, cot, p, W, H)
    kd, zd, cd = keys.cuda(), z.cuda(), cot.cuda()
    pd = p.cuda() if pad else None
    outs = []
    for _ in range(2):
        g_z, g_k, tag = run(kd, zd, cd, pd, W, H, C, tickets, mod.DEBUG_FORCE_SORTED_SEG | mod.DEBUG_FORCE_HOT)
        assert tag.startswith("slice_bwd_sorted2s"), tag
        outs.append((g_z, g_k))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "not bitwise reproducible"
    assert per_channel_err(outs[0][0], gz_ref, H * C) <= 1e-4
    assert relerr(outs[0][1], gk_ref) <= 1e-4


def test_sorted_segments_2d_against_the_one_workgroup_per_plane_kernel():
    """The zoo's 16^2 C16 head at B8 H16 N4096 (the default takes sorted segments): against the 1024-thread sorted kernel with
    channel groups (csrc/ct_raster_sorted.h, CLOUDCT-independent: forced by CT_DEBUG_FORCE_SORTED) to 1e-5."""
    mod, lib = _lib()
    B, H, C, N, W = 8, 16, 16, 4096, (16, 16)
    torch.manual_seed(5)
    keys = torch.tanh(torch.randn(B, H * 2, N, device="cuda"))
    z = torch.randn(B, H * C, *W, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    a = run(keys, z, cot, None, W, H, C, True, 0)
    b = run(keys, z, cot, None, W, H, C, True, mod.DEBUG_FORCE_SORTED)
    assert a[2].startswith("slice_bwd_sorted2s_segments"), a[2]
    assert b[2].startswith("slice_bwd_sorted_groups"), b[2]
    assert per_channel_err(a[0], b[0], H * C) <= 1e-5
    assert relerr(a[1], b[1]) <= 1e-5


def test_sorted_segments_2d_with_non_finite_channels():
    """The 2D form takes the same IEEE float-atomic pass for a channel with inf / NaN (ct_raster_hot.h: scatter_float_channel)."""
    mod, lib = _lib()
    W, B, H, C, N = (16, 16), 1, 2, 8, 1024
    g = torch.Generator().manual_seed(7)
    keys = torch.tanh(torch.randn(B, H * 2, N, generator=g))
    grid = torch.randn(B, H * C, *W, generator=g)
    cot = torch.randn(B, H * C, N, generator=g)
    cot[0, 1, 7] = float("inf")
    cot[0, C + 6] *= 1e30
    ref, _ = oracle(keys, grid, cot, None, W, H)
    got, _, tag = run(keys.cuda(), grid.cuda(), cot.cuda(), None, W, H, C, False, mod.DEBUG_FORCE_SORTED_SEG | mod.DEBUG_FORCE_HOT)
    assert tag.startswith("slice_bwd_sorted2s"), tag
    got = got.cpu()
    for ch in range(H * C):
        a, r = got[0, ch].double(), ref[0, ch].double()
        fin = torch.isfinite(r)
        assert torch.equal(torch.isfinite(a), fin), ch
        assert float((a[fin] - r[fin]).abs().max()) <= 1e-4 * float(r[fin].abs().max()), ch

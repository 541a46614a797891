"""Sorted-plane raster backward (csrc/ct_raster_sorted.h): ct_plane_sort's record and the Slice backward that walks it.

Reference semantics: layers/cloud_transform.py:164-173 (Slice.forward gathers the 2^d corners) and :216-221 (its backward is
the scatter-add of the corner products); the key cotangent follows layers/cloud_transform.py:91-94 through the weights.
Bars: g_grid within 1e-4 of EACH CHANNEL's own max (the item sums are rounded once to a per-channel fixed-point quantum), g_keys
within 1e-4; the record is a pure function of the keys (a deterministic counting sort), so everything here is bitwise reproducible."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

K_MAX_ITEMS = 2048


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


def plane_sort(keys, H, W):
    """-> (record bytes as a (B*H, stride) uint8 array on the host, stride)"""
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    B, _, N = keys.shape
    Wa = mod.int_array(list(W))
    n = lib.ct_plane_sort_bytes(B, H, N, 2, Wa)
    assert n > 0 and n % (B * H) == 0
    rec = torch.zeros(n, device="cuda", dtype=torch.uint8)
    mod.check(lib.ct_plane_sort(_ptr(keys), _ptr(rec), n, B, H, N, 2, Wa, _stream()), "ct_plane_sort")
    torch.cuda.synchronize()
    return rec, n // (B * H)


def parse_record(row, N):
    """one plane's record -> dict (layout: csrc/ct_raster_sorted.h, 'The sorted plane as a RECORD')"""
    row = row.cpu().numpy()
    hdr = row[:16].view(np.uint32)
    ab = row[16:16 + 8 * N].view(np.float32).reshape(N, 2)
    rk = row[16 + 8 * N:16 + 10 * N].view(np.uint16)
    items = row[16 + 10 * N:16 + 10 * N + 4 * K_MAX_ITEMS].view(np.uint32)
    return dict(nitems=int(hdr[0]), K=int(hdr[1]), ab=ab, rank=(rk & 0x1fff).astype(np.int64), inside_x=(rk & 0x4000) != 0,
                inside_y=(rk & 0x8000) != 0, items=items[:int(hdr[0])])


@pytest.mark.parametrize("shape", [(2, 3, 1024, (32, 32)), (1, 2, 4096, (32, 32)), (1, 2, 516, (16, 24)), (1, 1, 4096, (4, 4)),
                                   (1, 2, 2052, (32, 32))], ids=str)
def test_plane_sort_record_is_a_deterministic_counting_sort_of_the_base_cells(shape):
    """Ranks: a permutation that orders the points by base cell and, inside a cell, by (wave of the owning thread, which of the
    thread's four points) — point p belongs to thread p // 4; lanes that hit one cell in the SAME instruction are ranked in
    the order the LDS serves them, which is fixed but not the lane order (so the sort is reproducible, not stable: measured);
    weights in sorted order bit for bit the oracle's fractional parts;
    items: runs of <= 4 consecutive entries of one cell that cover the list exactly once; K = the largest number of
    contributions to a grid cell; the clamp masks of torch.clamp's backward."""
    B, H, N, W = shape
    g = torch.Generator().manual_seed(11)
    keys = torch.tanh(torch.randn(B, H * 2, N, generator=g) * 1.5)
    keys[0, 0, :8] = torch.tensor([-1.0, 1.0, -0.99999994, 0.99999994, 0.0, 0.5, -2.0, 3.0])     # clamp edges
    keys[0, 1, N // 2:N // 2 + 64] = keys[0, 1, :64]                                                # duplicates share cells
    keys[0, 0, N // 2:N // 2 + 64] = keys[0, 0, :64]
    rec, stride = plane_sort(keys.cuda(), H, W)
    rec = rec.reshape(B * H, stride)
    lo, hi = np.float32(-0.99999988), np.float32(0.99999988)
    for bh in range(B * H):
        b, h = divmod(bh, H)
        k = keys[b, 2 * h:2 * h + 2].numpy()
        kc = np.minimum(np.maximum(k, lo), hi)
        s = (kc + np.float32(1)) * np.float32([[(W[0] - 1) * 0.5], [(W[1] - 1) * 0.5]])
        fl = np.floor(s)
        base = np.minimum(fl[0].astype(np.int64), W[0] - 2) * W[1] + np.minimum(fl[1].astype(np.int64), W[1] - 2)
        r = parse_record(rec[bh], N)
        assert np.array_equal(np.sort(r["rank"]), np.arange(N)), "not a permutation"
        order = np.argsort(r["rank"])                  # order[i] = the point at sorted position i
        p = np.arange(N)
        sub = (p // 256) * 4 + (p % 4)                 # (wave, point of the thread): 64 lanes x 4 points per wave
        key = base * 64 + sub
        assert np.all(np.diff(key[order]) >= 0), "not ordered by (base cell, wave, point of the thread)"
        want_rank = r["rank"]
        w1 = (s - fl).astype(np.float32)
        assert np.array_equal(r["ab"][want_rank, 0].view(np.uint32), w1[0].view(np.uint32))
        assert np.array_equal(r["ab"][want_rank, 1].view(np.uint32), w1[1].view(np.uint32))
        assert np.array_equal(r["inside_x"], (k[0] >= lo) & (k[0] <= hi)) and np.array_equal(r["inside_y"], (k[1] >= lo) & (k[1] <= hi))
        # items
        first, n, cell = r["items"] & 0x1fff, ((r["items"] >> 13) & 3) + 1, r["items"] >> 16
        sorted_base = base[order]
        pos = 0
        for f, m, c in zip(first, n, cell):
            assert f == pos and 1 <= m <= 4 and np.all(sorted_base[f:f + m] == c)
            pos += m
            assert m == 4 or pos == N or sorted_base[pos] != c, "an item ends early inside its cell"
        assert pos == N
        cnt = np.bincount(base, minlength=W[0] * W[1]).reshape(W)
        contrib = cnt.copy()
        contrib[1:, :] += cnt[:-1, :]
        contrib[:, 1:] += cnt[:, :-1]
        contrib[1:, 1:] += cnt[:-1, :-1]
        assert r["K"] == contrib.max()
    # a pure function of the keys
    rec2, _ = plane_sort(keys.cuda(), H, W)
    assert torch.equal(rec.reshape(-1), rec2)


def test_plane_sort_refuses_layouts_without_a_sorted_form():
    mod, lib = _lib()
    assert lib.ct_plane_sort_bytes(2, 4, 1024, 3, mod.int_array([8, 8, 8])) == 0         # 3D
    assert lib.ct_plane_sort_bytes(2, 4, 8192, 2, mod.int_array([32, 32])) == 0          # more points than a workgroup owns
    assert lib.ct_plane_sort_bytes(2, 4, 1022, 2, mod.int_array([32, 32])) == 0          # rows not 16-byte addressable
    assert lib.ct_plane_sort_bytes(2, 4, 4096, 2, mod.int_array([64, 64])) == 0          # N / 4 + 3 G / 4 items > 2 per thread
    assert lib.ct_plane_sort_bytes(2, 4, 1024, 2, mod.int_array([32, 32])) > 0


SMALL = [
    # B, H, C, N, W, pad, duplicated points
    (2, 3, 8, 1024, (32, 32), False, False),
    (1, 2, 16, 4096, (32, 32), False, False),
    (2, 2, 12, 516, (16, 24), True, False),        # non-square, padding mask, ragged last quad of threads
    (1, 2, 8, 256, (8, 8), False, True),
    (1, 1, 20, 2048, (16, 16), True, False),
    (1, 2, 8, 2052, (32, 32), False, False),
    (1, 2, 8, 4096, (32, 32), False, True),         # heavy duplicates: cells with many items
    (1, 1, 8, 4096, (4, 4), False, False),          # 9 base cells: the items of one cell fill whole waves
]


@pytest.mark.parametrize("cfg", SMALL, ids=str)
@pytest.mark.parametrize("record", [False, True], ids=["sort_inside", "record"])
def test_sorted_slice_backward_against_the_oracle(cfg, record, flags):
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    B, H, C, N, W, pad, dup = cfg
    dim = 2
    torch.manual_seed(7)
    keys = torch.tanh(torch.randn(B, H * dim, N) * (0.3 if dup else 1.0))
    if dup:
        keys[:, :, N // 2:] = keys[:, :, :N // 2]
    if W == (4, 4):
        keys = keys * 0.2
    z = torch.randn(B, H * C, *W)
    cot = torch.randn(B, H * C, N)
    p = (torch.rand(B, N) > 0.2).float() if pad else None
    k = keys.clone().requires_grad_(True)
    zz = z.clone().requires_grad_(True)
    lc, idx = R.positions(k, list(W), H, dim)
    R.slice_(lc, idx, zz, p, list(W), H, dim).backward(cot)
    kd, zd, cd = keys.cuda(), z.cuda(), cot.cuda()
    pd = p.cuda() if pad else None
    Wa = mod.int_array(list(W))
    rec = None
    if record:
        rec, _ = plane_sort(kd, H, W)
    outs = []
    for _ in range(2):
        g_z, g_k = torch.full_like(zd, float("nan")), torch.full_like(kd, float("nan"))
        flags(mod.DEBUG_FORCE_SORTED | mod.DEBUG_FORCE_HOT)
        mod.check(lib.ct_slice_bwd_ps(_ptr(kd), _ptr(zd), _ptr(pd), mod.PAD_F32 if pad else 0, _ptr(cd), _ptr(g_z), _ptr(g_k),
                                      None, 0, None, _ptr(rec), B, H, C, N, dim, Wa, _stream()), "ct_slice_bwd_ps")
        tag = lib.ct_debug_last_launch().decode()
        flags(0)
        assert tag == ("slice_bwd_presorted" if record else "slice_bwd_sorted"), tag       # (no workspace: one workgroup per plane)
        outs.append((g_z, g_k))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "not bitwise reproducible"
    assert per_channel_err(outs[0][0], zz.grad, H * C) <= 1e-4
    assert relerr(outs[0][1], k.grad) <= 1e-4


@pytest.mark.parametrize("cfg", SMALL, ids=str)
def test_sorted_splat_sum_forward_against_the_oracle(cfg, flags):
    """Splat(reduce=sum) forward = the scatter-add side of the sorted kernel alone (no conv tile, no key cotangent):
    layers/cloud_transform.py:164-173 with scatter_add_."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    B, H, C, N, W, pad, dup = cfg
    dim = 2
    torch.manual_seed(9)
    keys = torch.tanh(torch.randn(B, H * dim, N) * (0.3 if dup else 1.0))
    if dup:
        keys[:, :, N // 2:] = keys[:, :, :N // 2]
    if W == (4, 4):
        keys = keys * 0.2
    feat = torch.randn(B, H * C, N)
    p = (torch.rand(B, N) > 0.2).float() if pad else None
    lc, idx = R.positions(keys, list(W), H, dim)
    ref = R.splat(lc, idx, feat, p, list(W), H, dim, "sum")
    outs = []
    for _ in range(2):
        flags(mod.DEBUG_FORCE_SORTED | mod.DEBUG_FORCE_HOT)
        z = ops.splat_keys(keys.cuda(), feat.cuda(), p.cuda() if pad else None, list(W), H, dim, "sum")
        tag = lib.ct_debug_last_launch().decode()
        flags(0)
        assert tag == "scatter_add_sorted", tag
        outs.append(z)
    assert torch.equal(outs[0], outs[1]), "not bitwise reproducible"
    assert per_channel_err(outs[0], ref, H * C) <= 1e-4


def test_record_and_inside_sort_agree_bit_for_bit_and_with_the_scatter_form(flags):
    """B4 H64 (256 planes: the sorted form is the default) N4096 C16 32^2: the three forms of Slice backward on one input."""
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    B, H, C, N, W, dim = 4, 64, 16, 4096, 32, 2
    torch.manual_seed(3)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    z = torch.randn(B, H * C, W, W, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    Wa = mod.int_array([W, W])
    rec, _ = plane_sort(keys, H, (W, W))
    nws = lib.ct_slice_bwd_workspace_bytes(B, H, C, N, dim, Wa)
    ws = torch.empty(max(nws, 16), device="cuda", dtype=torch.uint8)
    res = {}
    for name, fl, r in (("scatter", mod.DEBUG_NO_SORTED, None), ("inside", 0, None), ("record", 0, rec)):
        g_z, g_k = torch.empty_like(z), torch.empty_like(keys)
        flags(fl)
        mod.check(lib.ct_slice_bwd_ps(_ptr(keys), _ptr(z), None, 0, _ptr(cot), _ptr(g_z), _ptr(g_k), _ptr(ws), nws, None, _ptr(r),
                                      B, H, C, N, dim, Wa, _stream()), name)
        res[name] = (g_z, g_k, lib.ct_debug_last_launch().decode())
        flags(0)
    assert res["scatter"][2] == "slice_bwd_fused" and res["inside"][2] == "slice_bwd_sorted" and res["record"][2] == "slice_bwd_presorted"
    assert torch.equal(res["inside"][0], res["record"][0]) and torch.equal(res["inside"][1], res["record"][1])
    assert per_channel_err(res["inside"][0], res["scatter"][0], H * C) <= 1e-5
    assert relerr(res["inside"][1], res["scatter"][1]) <= 1e-5


@pytest.mark.parametrize("cfg", [(8, 16, 16, 4096, (16, 16)), (2, 16, 32, 2048, (32, 32)), (1, 3, 12, 1024, (16, 24))], ids=str)
def test_channel_groups_of_a_plane_on_several_workgroups(cfg, flags):
    """Few planes (the H16 blocks): a plane's channel groups are dealt to 2+ workgroups, each with its own sort, the partial
    g_keys added by the plane's last workgroup (tickets) or by a sum_parts launch — against the oracle, and bit for bit the
    same with and without the tickets (the partials are added in ascending order either way)."""
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    B, H, C, N, W = cfg
    dim = 2
    torch.manual_seed(17)
    keys = torch.tanh(torch.randn(B, H * dim, N))
    z = torch.randn(B, H * C, *W)
    cot = torch.randn(B, H * C, N)
    k = keys.clone().requires_grad_(True)
    zz = z.clone().requires_grad_(True)
    lc, idx = R.positions(k, list(W), H, dim)
    R.slice_(lc, idx, zz, None, list(W), H, dim).backward(cot)
    kd, zd, cd = keys.cuda(), z.cuda(), cot.cuda()
    Wa = mod.int_array(list(W))
    nws = max(lib.ct_slice_bwd_workspace_bytes(B, H, C, N, dim, Wa), 2 * keys.numel() * 4)
    ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
    outs = {}
    for tickets in (False, True):
        tk = torch.zeros(mod.TICKETS_BYTES // 4, device="cuda", dtype=torch.int32) if tickets else None
        g_z, g_k = torch.full_like(zd, float("nan")), torch.full_like(kd, float("nan"))
        flags(mod.DEBUG_FORCE_SORTED | mod.DEBUG_FORCE_HOT)
        mod.check(lib.ct_slice_bwd_ps(_ptr(kd), _ptr(zd), None, 0, _ptr(cd), _ptr(g_z), _ptr(g_k), _ptr(ws), nws, _ptr(tk), None,
                                      B, H, C, N, dim, Wa, _stream()), "ct_slice_bwd_ps")
        tag = lib.ct_debug_last_launch().decode()
        flags(0)
        assert tag == ("slice_bwd_sorted_groups+folded" if tickets else "slice_bwd_sorted_groups"), tag
        if tickets:
            assert int(tk.abs().sum()) == 0, "the tickets were not handed back as zeros"
        assert per_channel_err(g_z, zz.grad, H * C) <= 1e-4
        assert relerr(g_k, k.grad) <= 1e-4
        outs[tickets] = (g_z, g_k)
    assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1]), "tickets / sum_parts differ"


def test_sorted_slice_backward_with_non_finite_channels(flags):
    """A channel that holds inf / NaN (or would overflow the fixed-point bound) takes IEEE float atomics in the sorted form too;
    every other channel keeps its accuracy (here the channels are NOT paired: only the channel itself leaves the integers)."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    W, dim, B, H, C, N = (32, 32), 2, 1, 2, 8, 1024
    g = torch.Generator().manual_seed(7)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    grid = torch.randn(B, H * C, *W, generator=g)
    cot = torch.randn(B, H * C, N, generator=g)
    cot[0, 1, 7] = float("inf")
    cot[0, 4, 100] = float("nan")
    cot[0, C + 6] *= 1e30
    lc, idx = R.positions(keys, list(W), H, dim)
    gr = grid.clone().requires_grad_(True)
    R.slice_(lc, idx, gr, None, list(W), H, dim).backward(cot)
    ref = gr.grad
    flags(mod.DEBUG_FORCE_HOT | mod.DEBUG_FORCE_SORTED)
    gk = grid.cuda().requires_grad_(True)
    ops.slice_keys(keys.cuda(), gk, None, list(W), H, dim).backward(cot.cuda())
    tag = lib.ct_debug_last_launch().decode()
    flags(0)
    assert tag.startswith("slice_bwd_sorted"), tag          # (two channel-group workgroups per plane when forced on a small shape)
    got = gk.grad.cpu()
    for ch in range(H * C):
        a, r = got[0, ch].double(), ref[0, ch].double()
        if ch in (1, 4):
            assert torch.equal(torch.isfinite(a), torch.isfinite(r)), ch
            fin = torch.isfinite(r)
            assert float((a[fin] - r[fin]).abs().max()) <= 1e-4 * float(r[fin].abs().max()), ch
        else:
            assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()), ch


@pytest.mark.parametrize("cfg", [(1, 2, 8, 8192, (32, 32)), (16, 16, 8, 512, (8, 8, 8)), (2, 2, 8, 1024, (8, 8, 8))], ids=str)
def test_in_place_key_accumulation_with_exact_ties_in_the_through_memory_forms(cfg, flags):
    """ct_splat_bwd_ex(CT_BWD_ACCUMULATE_KEYS) on a cloud with duplicated points, in the forms of the hot
    Splat(max) backward whose g_keys sums go through memory (2D beyond 4096 points per workgroup, every 3D call) with ONE chunk
    group: the single-winner redo of a tied plane must start from the INCOMING cotangent, not from incoming + the optimistic
    pass.  Compared with the plain call (no accumulation) on the same input."""
    from cloud_transformers_amd.ops import _ptr, _stream
    mod, lib = _lib()
    B, H, C, N, W = cfg
    dim = len(W)
    g = torch.Generator().manual_seed(23)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    keys[:, :, N // 2:] = keys[:, :, :N // 2]          # every point twice: every winning product ties
    feat[:, :, N // 2:] = feat[:, :, :N // 2]
    keys, feat = keys.cuda(), feat.cuda()
    gz = torch.randn(B, H * C, *W, generator=g).cuda()
    base = torch.randn(B, H * dim, N, generator=g).cuda()
    Wa = mod.int_array(list(W))
    z = torch.empty(B, H * C, *W, device="cuda")
    mod.check(lib.ct_splat_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(z), B, H, C, N, dim, Wa, 0, _stream()), "fwd")
    half = N // 2
    for dbg in (mod.DEBUG_FORCE_HOT, 0):
        flags(dbg)
        n0 = lib.ct_splat_bwd_workspace_bytes(B, H, C, N, dim, Wa, 0)
        ws0 = torch.empty(max(n0, 16), device="cuda", dtype=torch.uint8)
        gf_a, gk_plain = torch.empty_like(feat), torch.empty_like(keys)
        mod.check(lib.ct_splat_bwd(_ptr(keys), _ptr(feat), None, 0, _ptr(z), _ptr(gz), _ptr(gf_a), _ptr(gk_plain), _ptr(ws0), n0,
                                   B, H, C, N, dim, Wa, 0, _stream()), "bwd")
        n1 = lib.ct_splat_bwd_ex_workspace_bytes(B, H, C, N, dim, Wa, 0, mod.BWD_ACCUMULATE_KEYS)
        ws1 = torch.empty(n1, device="cuda", dtype=torch.uint8)
        gf_b, gk_acc = torch.empty_like(feat), base.clone()
        mod.check(lib.ct_splat_bwd_ex(_ptr(keys), _ptr(feat), None, 0, _ptr(z), _ptr(gz), _ptr(gf_b), _ptr(gk_acc), _ptr(ws1), n1,
                                      B, H, C, N, dim, Wa, 0, mod.BWD_ACCUMULATE_KEYS, _stream()), "bwd_ex")
        tag = lib.ct_debug_last_launch().decode()
        flags(0)
        # which copy of a duplicated point wins is not pinned between two launches: the PAIR's gradient mass is
        pair = lambda t: t[..., :half] + t[..., half:]
        assert relerr(pair(gf_b), pair(gf_a)) <= 1e-5, tag
        assert relerr(pair(gk_acc - base), pair(gk_plain)) <= 1e-4, tag

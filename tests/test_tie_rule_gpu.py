"""Exact ties in Splat(max) backward: ONE winner per (cell, channel) in every kernel family — never a split award.

torch_scatter.scatter_max's backward (layers/cloud_transform.py:164-173 through torch_scatter) gives the cotangent of a cell to the
single arg-max element it recorded; which of several bit-equal contributions that is differs between its CPU and CUDA kernels.
The rule here, per family (HISTORY.md §2): the hot kernels repair a single chance tie in favour of the lowest point index and
redo anything beyond it with claims; there, and in the generic and quad kernels, the contribution whose compare-and-swap reaches
the cell's word first wins (one winner, identity unspecified); the banded kernels award the lowest point index.

Construction: two identical points alone in their neighbourhood, one cotangent value per corner cell.  The four corner cells see
exactly the two tied contributions each, so the pair's gradients must be a PARTITION of the four corner terms g_z * weight."""
import itertools

import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _lib():
    from cloud_transformers_amd import _lib
    return _lib, _lib.load()


CASES = [
    # name, debug flags, dim, W, C, N, lowest index must win
    ("hot_2d_register_form", "FORCE_HOT", 2, 32, 8, 1024, False),
    ("hot_2d_through_memory", "FORCE_HOT", 2, 32, 8, 8192, False),
    ("hot_3d", "FORCE_HOT", 3, 8, 8, 1024, False),
    ("generic_2d", "NO_HOT", 2, 32, 8, 1024, False),
    ("generic_3d", "NO_HOT", 3, 8, 8, 1000, False),
    ("banded_2d", "FORCE_BAND", 2, 32, 4, 1024, True),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_one_winner_per_cell_and_channel(case):
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    name, flag, dim, W, C, N, lowest = case
    B, H = 2, 16
    g = torch.Generator().manual_seed(41)
    Ws = [W] * dim
    # the cloud sits in the lower half of the grid; the tied pair alone near the upper corner
    keys = torch.rand(B, H * dim, N, generator=g) * 0.8 - 0.9
    p0, p1 = 5, N - 7
    spot = torch.rand(B, H * dim, generator=g) * 0.1 + 0.7
    keys[:, :, p0] = spot
    keys[:, :, p1] = spot
    feat = torch.rand(B, H * C, N, generator=g) + 0.5
    feat[:, :, p1] = feat[:, :, p0]
    gz = torch.randn(B, H * C, *Ws, generator=g)
    lc, idx = R.positions(keys, Ws, H, dim)                          # (B, H, V, N): corner weights / cells
    V = 1 << dim
    kd, fd = keys.cuda().requires_grad_(True), feat.cuda().requires_grad_(True)
    lib.ct_debug_set_flags(getattr(mod, "DEBUG_" + flag))
    try:
        z = ops.splat_keys(kd, fd, None, Ws, H, dim, "max")
        z.backward(gz.cuda())
        tag = lib.ct_debug_last_launch().decode()
    finally:
        lib.ct_debug_set_flags(0)
    want = {"FORCE_HOT": ("hot",), "NO_HOT": ("generic", "quad", "whole"), "FORCE_BAND": ("band",)}[flag]
    assert any(w in tag for w in want) and ("hot" in tag) == (flag == "FORCE_HOT"), (name, tag)
    gf = fd.grad.cpu()
    G = W ** dim
    for b in range(B):
        for h in range(H):
            w = lc[b, h, :, p0]                                      # the pair's corner weights (identical for both)
            cells = idx[b, h, :, p0]
            for c in range(C):
                terms = gz[b, h * C + c].reshape(G)[cells] * w       # g_z * weight per corner
                a, bb = float(gf[b, h * C + c, p0]), float(gf[b, h * C + c, p1])
                scale = float(terms.abs().sum()) + 1e-12
                sums = {S: float(sum(terms[v] for v in S)) for k in range(V + 1) for S in itertools.combinations(range(V), k)}
                full = sums[tuple(range(V))]
                assert abs(a + bb - full) <= 1e-5 * scale, (name, tag, b, h, c)
                assert any(abs(a - s) <= 1e-5 * scale for s in sums.values()), \
                    "%s (%s): plane (%d,%d) channel %d: the pair's award %.6g / %.6g splits a corner term" % (name, tag, b, h, c, a, bb)
                if lowest:
                    assert abs(a - full) <= 1e-5 * scale and abs(bb) <= 1e-5 * scale, (name, tag, "the lower point index wins every corner")


@pytest.mark.parametrize("flag", ["FORCE_HOT", "NO_HOT"])
def test_a_zero_valued_candidate_does_not_win_a_cell_at_the_zero_floor(flag):
    """HISTORY.md §2, "Zero-valued candidates": a point exactly on a cell boundary (W = 9: (key + 1) * 4 is exact) sends a
    product of exactly 0 into the far cells; where those stay at the zero floor torch_scatter would route their cotangent to the
    point (0 == 0), torch's amax half of it, the kernels here none.  The expected values are the oracle's with the cotangent
    of the far cells removed."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    B, H, C, N, W, dim = 1, 2, 4, 8, 9, 2
    keys = torch.full((B, H * dim, N), -0.9)
    keys[:, 0::2, 3] = -0.5          # x of point 3: scaled coordinate exactly 2.0 -> weight 0 towards row 3
    keys[:, 1::2, 3] = 0.3
    feat = torch.rand(B, H * C, N) + 0.5
    gz = torch.randn(B, H * C, W, W)
    lc, idx = R.positions(keys, [W, W], H, dim)
    assert float(lc[0, 0, :, 3].min()) == 0.0
    k = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    lcr, idxr = R.positions(k, [W, W], H, dim)
    z = R.splat(lcr, idxr, f, None, [W, W], H, dim, "max")
    gz_near = gz.clone().reshape(B, H * C, W * W)
    far = idx[0, 0, :, 3][lc[0, 0, :, 3] == 0]                    # the cells the zero products go to (empty otherwise)
    assert float(z.detach().reshape(B, H * C, -1)[:, :, far].abs().max()) == 0.0
    gz_near[:, :, far] = 0.0
    z.backward(gz_near.reshape_as(gz))
    kd, fd = keys.cuda().requires_grad_(True), feat.cuda().requires_grad_(True)
    lib.ct_debug_set_flags(getattr(mod, "DEBUG_" + flag))
    try:
        ops.splat_keys(kd, fd, None, [W, W], H, dim, "max").backward(gz.cuda())
    finally:
        lib.ct_debug_set_flags(0)
    assert float((kd.grad.cpu() - k.grad).abs().max()) <= 1e-5 * float(k.grad.abs().max())
    assert float((fd.grad.cpu() - f.grad).abs().max()) <= 1e-5 * float(f.grad.abs().max())


def _node_key(W):
    """a float32 key whose scaled coordinate (key + 1) * ((W - 1) / 2), rounded as ct_axis rounds it, is an exact integer j >= 1:
    the point then sits on a grid node and three of its four corner weights are exactly 0"""
    import numpy as np
    hw = np.float32((W - 1) / 2.0)
    for j in range(2, W - 2):
        k = np.float32(2.0 * j / (W - 1) - 1.0)
        for _ in range(8):
            s = np.float32(np.float32(k + np.float32(1.0)) * hw)
            if float(s) == float(j):
                return float(k), j
            k = np.nextafter(k, np.float32(1.0 if s < j else -1.0), dtype=np.float32)
    raise AssertionError("no node key for W = %d" % W)


@pytest.mark.parametrize("shape", [(32, 16, 4096, 2), (32, 16, 2048, 1), (16, 16, 4096, 1), (20, 8, 2048, 1), (32, 16, 4096, -2)],
                         ids=["32x32_two_quads", "32x32_one_quad", "16x16", "generic_width", "32x32_padding_mask"])
def test_a_single_tie_in_one_group_of_one_plane(shape):
    """One surplus match in one four-channel group of one plane (what a random B8 H64 workload holds about one time in three).
    The hot 2D kernel redoes that group alone (splat_bwd_quad<.., DELTA>) or, built with -DCT_TIE_FIX=1, finds the cell from its
    cell sums and repairs it alone (splat_bwd_fix_one_tie: the lower point index keeps the award).  Two points on the same grid
    node (one non-zero corner weight), equal and dominant in ONE channel: exactly one tie.  Checked for either build: one of the
    two keeps the cell's cotangent, the other gets exactly nothing from it (features and keys), and every element outside the
    pair equals the same launch without the second point's feature (no tie at all)."""
    from cloud_transformers_amd import ops
    mod, lib = _lib()
    W, C, N, quads = shape
    B, H, dim = 4, 64, 2                     # a plane per workgroup: the register form with the per-group counters
    pad = None
    if quads < 0:                            # with a padding mask (float32 0 / 1 per point: the kernels' HAS_PAD instantiations)
        pad = (torch.rand(B, N, generator=torch.Generator().manual_seed(3)) > 0.1).float()
        pad[:, 37] = 1.0
        pad[:, N - 5] = 1.0
    g = torch.Generator().manual_seed(7)
    k_node, j = _node_key(W)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    gz = torch.randn(B, H * C, W, W, generator=g)
    p0, p1 = 37, N - 5                       # different threads, different quads
    b0, h0, c0 = 1, 3, C - 2                 # (the last four-channel group of its chunk)
    keys[b0, h0 * dim:(h0 + 1) * dim, p0] = k_node
    keys[b0, h0 * dim:(h0 + 1) * dim, p1] = k_node
    feat[b0, h0 * C:(h0 + 1) * C, p1] = -1.0              # never above the zero floor ...
    feat[b0, h0 * C + c0, p0] = 100.0
    feat_untied = feat.clone()
    feat[b0, h0 * C + c0, p1] = 100.0                      # ... except here: bit-equal to p0's product
    lc, idx = R.positions(keys, [W, W], H, dim)
    assert sorted(lc[b0, h0, :, p0].tolist())[:3] == [0.0, 0.0, 0.0]
    cell = int(idx[b0, h0, :, p0][lc[b0, h0, :, p0] > 0][0])
    assert cell == j * W + j

    def run(f):
        kd, fd = keys.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
        lib.ct_debug_set_flags(mod.DEBUG_FORCE_HOT)
        try:
            ops.splat_keys(kd, fd, None if pad is None else pad.cuda(), [W, W], H, dim, "max").backward(gz.cuda())
            tag = lib.ct_debug_last_launch().decode()
        finally:
            lib.ct_debug_set_flags(0)
        assert "hot" in tag, tag
        return kd.grad.cpu(), fd.grad.cpu()

    gk, gf = run(feat)
    gk0, gf0 = run(feat_untied)
    award = float(gz[b0, h0 * C + c0].reshape(-1)[cell])
    a0, a1 = float(gf[b0, h0 * C + c0, p0]), float(gf[b0, h0 * C + c0, p1])
    assert (a0 == pytest.approx(award, rel=1e-6) and a1 == 0.0) or (a1 == pytest.approx(award, rel=1e-6) and a0 == 0.0), (a0, a1, award)
    if a1 == 0.0:          # p0 kept it: p1 (whose other features never pass the zero floor) ends with exactly nothing
        assert float(gf[b0, h0 * C:(h0 + 1) * C, p1].abs().max()) == 0.0
        assert float(gk[b0, h0 * dim:(h0 + 1) * dim, p1].abs().max()) == 0.0
        assert torch.equal(gf, gf0)
        assert torch.equal(gk, gk0)
    else:                  # p1 kept it: the pair's rows differ from the untied launch, nothing else does
        keep = torch.ones(N, dtype=torch.bool)
        keep[p0] = keep[p1] = False
        assert torch.equal(gf[:, :, keep], gf0[:, :, keep])
        assert torch.equal(gk[:, :, keep], gk0[:, :, keep])
        lost = float(gk0[b0, h0 * dim:(h0 + 1) * dim, p0].abs().max())
        assert float((gk[b0, h0 * dim:(h0 + 1) * dim, p0] + gk[b0, h0 * dim:(h0 + 1) * dim, p1]
                      - gk0[b0, h0 * dim:(h0 + 1) * dim, p0]).abs().max()) <= 1e-5 * lost


MEM_FORMS = [
    # name, dim, W, C, N, B, H, forced segments, tickets
    ("segments_2d", 2, 16, 16, 4096, 2, 4, 4, True),
    ("segments_3d", 3, 8, 32, 4096, 2, 4, 4, True),
    ("chunk_groups_3d_folded", 3, 8, 32, 4096, 8, 16, 0, True),
    ("chunk_groups_3d_two_launches", 3, 8, 32, 4096, 8, 16, 0, False),
    ("planes_3d", 3, 8, 16, 2048, 4, 64, 0, False),
    ("through_memory_2d", 2, 32, 8, 16384, 2, 128, 0, False),
    ("segments_2d_padding_mask", 2, 16, 16, 4096, 2, 4, 4, True),
    ("planes_3d_padding_mask", 3, 8, 16, 2048, 4, 64, 0, False),
]


@pytest.mark.parametrize("form", MEM_FORMS, ids=[f[0] for f in MEM_FORMS])
def test_a_single_tie_is_repaired_in_the_rows_it_touched(form):
    """The forms of the hot Splat(max) backward whose key cotangents are in memory when a tie is known (3D, point segments, N
    beyond the register forms) used to redo the whole pass of the workgroup — or the whole plane by its last workgroup — with
    claims: twice the time for one tie.  One surplus match is now repaired through memory (ct_raster_hot3d.h: splat_bwd_fix_mem):
    the lower point index keeps the award.  Same construction as the register form's test; the launch with the second point's
    feature removed (no tie) is the expected result: g_feat bit for bit, g_keys to rounding (the award is taken back by
    subtraction from the stored sum)."""
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.ops import _ptr, _stream
    lib = _lib.load()
    name, dim, W, C, N, B, H, nseg, use_tickets = form
    Wl = [W] * dim
    g = torch.Generator().manual_seed(13)
    k_node, j = _node_key(W)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    cot = torch.randn(B, H * C, *Wl, generator=g)
    add = torch.randn(B, H * dim, N, generator=g)
    p0, p1 = 37, N - 5                      # (different segments)
    b0, h0, c0 = B - 1, H // 2, C - 3
    keys[b0, h0 * dim:(h0 + 1) * dim, p0] = k_node
    keys[b0, h0 * dim:(h0 + 1) * dim, p1] = k_node
    feat[b0, h0 * C:(h0 + 1) * C, p1] = -1.0
    feat[b0, h0 * C + c0, p0] = 100.0
    feat_untied = feat.clone()
    feat[b0, h0 * C + c0, p1] = 100.0
    kd, cd, addd = keys.cuda(), cot.cuda(), add.cuda()
    padd, pad_code = None, 0
    if name.endswith("padding_mask"):        # (float32 0 / 1 per point: the kernels' HAS_PAD instantiations)
        pad = (torch.rand(B, N, generator=g) > 0.1).float()
        pad[:, p0] = 1.0
        pad[:, p1] = 1.0
        padd, pad_code = pad.cuda(), _lib.PAD_F32
    Wa = _lib.int_array(Wl)
    nws = lib.ct_splat_bwd_ex_workspace_bytes(B, H, C, N, dim, Wa, 0, 1)
    ws = torch.empty(max(nws, 16), device="cuda", dtype=torch.uint8)
    tickets = torch.zeros(_lib.TICKETS_BYTES // 4, device="cuda", dtype=torch.int32) if use_tickets else None

    def run(f):
        fd = f.cuda()
        z = torch.empty(B, H * C, *Wl, device="cuda")
        _lib.check(lib.ct_splat_fwd(_ptr(kd), _ptr(fd), _ptr(padd), pad_code, _ptr(z), B, H, C, N, dim, Wa, 0, _stream()), "fwd")
        g_feat = torch.full_like(fd, float("nan"))
        g_keys = torch.full_like(kd, float("nan"))
        lib.ct_debug_set_flags(_lib.DEBUG_FORCE_HOT)
        lib.ct_debug_set_nseg(nseg)
        try:
            _lib.check(lib.ct_splat_bwd_tk(_ptr(kd), _ptr(fd), _ptr(padd), pad_code, _ptr(z), _ptr(cd), _ptr(g_feat), _ptr(addd), _ptr(g_keys),
                                           _ptr(ws), nws, _ptr(tickets), B, H, C, N, dim, Wa, 0, _stream()), "bwd")
            torch.cuda.synchronize()
            tag = lib.ct_debug_last_launch().decode()
        finally:
            lib.ct_debug_set_flags(0)
            lib.ct_debug_set_nseg(0)
        return g_feat.cpu(), g_keys.cpu(), tag, z.cpu()

    gf, gk, tag, z = run(feat)
    gf0, gk0, tag0, z0 = run(feat_untied)
    assert "hot" in tag and tag == tag0, (tag, tag0)
    assert ("segments" in tag) == (nseg > 0), tag
    if use_tickets:
        assert int(tickets.abs().sum()) == 0
    assert torch.equal(z, z0)
    cell = j * sum(W ** e for e in range(dim))
    award = float(cot[b0, h0 * C + c0].reshape(-1)[cell])
    assert float(gf[b0, h0 * C + c0, p0]) == pytest.approx(award, rel=1e-6), (name, tag)
    assert float(gf[b0, h0 * C + c0, p1]) == 0.0, (name, tag)
    assert torch.equal(gf, gf0), (name, tag)
    assert float((gk - gk0).abs().max()) <= 1e-5 * float(gk0.abs().max()), (name, tag)

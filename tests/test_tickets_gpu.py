"""Arrival tickets (include/cloudct.h: ct_slice_bwd_tk / ct_splat_bwd_tk): the sums over the workgroups that share a (b,h)
plane — partial g_keys of the channel-chunk groups, partial g_grid tiles of the point segments — happen inside the
backward kernels instead of in sum_parts launches behind them, and the Splat(max) backward may deal a plane's POINTS to
several workgroups (no partial sums at all; the plane's exact-tie test then runs across them through the tickets).

The fold adds the partials in the order of the two-launch form, so the two forms must agree BIT FOR BIT on every
cotangent of Slice backward and on g_feat; the segmented Splat backward adds a point's channel chunks in one chain
instead of group by group, so its g_keys agree to rounding (held to 2e-6 of the tensor's max here, and to the oracle
below).  The two-launch form itself is held to the oracle by test_raster_gpu.py / test_headline_gpu.py.  A hand-off
between workgroups that went wrong (a partial read before it was visible, a stale line of the previous launch's
partials) shows as a wrong sum, so every case runs several launches with FRESH inputs on ONE workspace, and one case
runs two streams side by side so that workgroups of different launches interleave on the chip (uneven load).  The
ticket buffer must be all zero again after every launch."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, N, H, C, W, dim): the zoo's head shapes (model_zoo/s3dis/segmenter.py:28-45) at the segmenter's batch, at the
# completion decoder's (B2 N16384: point segments), at N2048, and small odd ones (planes not a multiple of 8: no XCD remap)
SHAPES = [
    (8, 4096, 16, 16, 64, 2), (8, 4096, 16, 16, 16, 3), (8, 4096, 16, 16, 16, 2), (8, 4096, 16, 32, 8, 3),
    (2, 16384, 16, 16, 64, 2), (2, 16384, 16, 16, 16, 3), (2, 16384, 16, 16, 16, 2), (2, 16384, 16, 32, 8, 3),
    (8, 2048, 16, 16, 16, 2), (8, 2048, 16, 32, 8, 3), (8, 2048, 16, 16, 16, 3),
    (3, 1024, 12, 16, 16, 2), (5, 2048, 7, 32, 8, 3), (4, 8192, 8, 16, 16, 2), (4, 8192, 8, 16, 16, 3),
]


def _steps(B, N, H, C, W, dim, seed):
    from cloud_transformers_amd.step import SplatSliceStep
    torch.manual_seed(seed)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    # the two steps share keys / feat / cot (read-only) and own their outputs, workspaces and tickets
    return (SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=True),
            SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=False))


def _refresh(step, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    step.keys.copy_(torch.tanh(torch.randn(step.keys.shape, device="cuda", generator=g)))
    step.feat.copy_(torch.randn(step.feat.shape, device="cuda", generator=g))
    step.cot.copy_(torch.randn(step.cot.shape, device="cuda", generator=g))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%dN%dH%dC%dW%dD%d" % s)
def test_folded_sums_equal_the_two_launch_form_bit_for_bit(shape):
    tk, two = _steps(*shape, seed=1)
    folded = False
    for it in range(4):
        _refresh(tk, 100 + it)             # (shared input tensors: `two` sees the same data)
        for st in (tk, two):
            st.splat_fwd()
            st.slice_fwd()
            st.slice_bwd()
        torch.cuda.synchronize()
        for name in ("z", "out", "g_z", "g_keys_buf"):            # (g_keys_buf: Slice's key cotangent, before Splat adds its own)
            a, b = getattr(tk, name), getattr(two, name)
            assert torch.equal(a, b), (name, it, float((a - b).abs().max()))
        tk.splat_bwd()
        two.splat_bwd()
        torch.cuda.synchronize()
        assert torch.equal(tk.g_feat, two.g_feat), (it, float((tk.g_feat - two.g_feat).abs().max()))
        a, b = tk.g_keys_out, two.g_keys_out
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (it, float((a - b).abs().max()))
        assert int(tk.tickets.abs().sum()) == 0, "tickets not reset"
        tags = tk.launch_tags()
        torch.cuda.synchronize()
        folded = folded or any("folded" in t or "segments" in t for t in tags.values())
        assert int(tk.tickets.abs().sum()) == 0
    B, N, H, C, W, dim = shape
    if B * H < 256 and C >= 16 and N <= 4096:     # (longer clouds: partials beyond the fold's size bound, two launches)
        assert folded, tags


def test_two_streams_side_by_side_keep_their_own_tickets():
    """Two launches in flight at once (a union block's two heads on their streams): each stream has its own ticket
    buffer, workgroups of the two launches interleave on the CUs, results still equal the two-launch form."""
    a_tk, a_two = _steps(8, 4096, 16, 16, 16, 2, seed=3)
    b_tk, b_two = _steps(8, 4096, 16, 32, 8, 3, seed=4)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for it in range(6):
        _refresh(a_tk, 200 + it)
        _refresh(b_tk, 300 + it)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            a_tk.run()
        with torch.cuda.stream(s2):
            b_tk.run()
        torch.cuda.synchronize()
        a_two.run()
        b_two.run()
        torch.cuda.synchronize()
        for t, r in ((a_tk, a_two), (b_tk, b_two)):
            for name in ("g_z", "g_feat"):
                assert torch.equal(getattr(t, name), getattr(r, name)), (name, it)
            assert float((t.g_keys_out - r.g_keys_out).abs().max()) <= 2e-6 * float(r.g_keys_out.abs().max()), it
            assert int(t.tickets.abs().sum()) == 0


def test_autograd_path_uses_the_tickets_and_matches_the_plain_entry_points():
    from cloud_transformers_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(5)
    B, N, H, C, W, dim = 8, 4096, 16, 16, 16, 3
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")

    def chain(on):
        old = ops.RASTER_TICKETS
        ops.RASTER_TICKETS = on
        try:
            k, f = keys.clone().requires_grad_(True), feat.clone().requires_grad_(True)
            z = ops.splat_keys(k, f, None, W, H, dim, "max")
            o = ops.slice_keys(k, z, None, W, H, dim)
            o.backward(cot)
            tag = lib.ct_debug_last_launch().decode()
            return k.grad, f.grad, tag
        finally:
            ops.RASTER_TICKETS = old

    gk1, gf1, tag1 = chain(True)
    gk0, gf0, tag0 = chain(False)
    assert ("folded" in tag1 or "segments" in tag1) and "folded" not in tag0 and "segments" not in tag0, (tag1, tag0)
    assert torch.equal(gf1, gf0)
    assert float((gk1 - gk0).abs().max()) <= 2e-6 * float(gk0.abs().max())


@pytest.mark.parametrize("shape", [(2, 4096, 4, 16, 16, 2), (2, 4096, 4, 32, 8, 3), (1, 16384, 2, 16, 16, 2)],
                         ids=lambda s: "B%dN%dH%dC%dW%dD%d" % s)
@pytest.mark.parametrize("dup", ["halves", "some"])
def test_ties_across_point_segments_route_to_a_single_winner(shape, dup):
    """Duplicated points whose copies sit in DIFFERENT segments of a plane: the segments cannot see the tie alone (each finds
    its own copy bit-equal to z); the plane's match count over all segments exceeds its non-zero cells, and the plane's last
    workgroup redoes it with single-winner claims from the incoming key cotangent.  Checked as in
    test_raster_gpu.py::test_exact_ties_route_to_a_single_winner: the copies' gradients add up to the gradient of the
    de-duplicated cloud (oracle), for g_feat and for the key cotangent on top of a non-trivial incoming one."""
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.ops import _ptr, _stream
    from oracle import ref_cpu as R
    lib = _lib.load()
    B, N, H, C, W, dim = shape
    Wl = [W] * dim
    g = torch.Generator().manual_seed(11)
    half = N // 2
    keys = torch.tanh(torch.randn(B, H * dim, half, generator=g))
    feat = torch.randn(B, H * C, half, generator=g)
    if dup == "halves":           # every point twice, the copy half a cloud away: always in another segment
        keys2, feat2 = keys.repeat(1, 1, 2), feat.repeat(1, 1, 2)
    else:                         # one plane of the batch with a few far-apart duplicates, the rest tie-free
        keys2 = torch.cat([keys, torch.tanh(torch.randn(B, H * dim, half, generator=g))], 2)
        feat2 = torch.cat([feat, torch.randn(B, H * C, half, generator=g)], 2)
        keys2[0, :dim, half:half + 64] = keys2[0, :dim, :64]
        feat2[0, :C, half:half + 64] = feat2[0, :C, :64]
    cot = torch.randn(B, H * C, *Wl, generator=g)
    add = torch.randn(B, H * dim, N, generator=g)
    kd, fd, cd, addd = keys2.cuda(), feat2.cuda(), cot.cuda(), add.cuda()
    z = torch.empty(B, H * C, *Wl, device="cuda")
    Wa = _lib.int_array(Wl)
    _lib.check(lib.ct_splat_fwd(_ptr(kd), _ptr(fd), None, 0, _ptr(z), B, H, C, N, dim, Wa, 0, _stream()), "fwd")
    nws = lib.ct_splat_bwd_ex_workspace_bytes(B, H, C, N, dim, Wa, 0, 1)
    ws = torch.empty(max(nws, 16), device="cuda", dtype=torch.uint8)
    tickets = torch.zeros(_lib.TICKETS_BYTES // 4, device="cuda", dtype=torch.int32)
    lib.ct_debug_set_flags(_lib.DEBUG_FORCE_HOT)
    lib.ct_debug_set_nseg(4)          # (automatic only for small 2D tiles: force the segmented form, 3D too)
    try:
        outs = []
        for tk in (tickets, None):       # segments + tickets, then the one-workgroup-per-plane form
            g_feat = torch.full_like(fd, float("nan"))
            g_keys = torch.full_like(kd, float("nan"))
            _lib.check(lib.ct_splat_bwd_tk(_ptr(kd), _ptr(fd), None, 0, _ptr(z), _ptr(cd), _ptr(g_feat), _ptr(addd), _ptr(g_keys),
                                           _ptr(ws), nws, _ptr(tk), B, H, C, N, dim, Wa, 0, _stream()), "bwd")
            torch.cuda.synchronize()
            outs.append((g_feat.cpu(), g_keys.cpu(), lib.ct_debug_last_launch().decode()))
    finally:
        lib.ct_debug_set_flags(0)
        lib.ct_debug_set_nseg(0)
    assert "segments" in outs[0][2], outs[0][2]
    assert "segments" not in outs[1][2], outs[1][2]
    assert int(tickets.abs().sum()) == 0
    if dup == "halves":
        k1 = keys.clone().requires_grad_(True)
        f1 = feat.clone().requires_grad_(True)
        lc, idx = R.positions(k1, Wl, H, dim)
        z1 = R.splat(lc, idx, f1, None, Wl, H, dim)
        assert torch.equal(z.cpu(), z1.detach())
        (z1 * cot).sum().backward()
        for gf, gk, _ in outs:
            e = (gf[..., :half] + gf[..., half:] - f1.grad).abs().max() / f1.grad.abs().max()
            assert float(e) < 1e-5, float(e)
            gk = gk - add
            e = (gk[..., :half] + gk[..., half:] - k1.grad).abs().max() / k1.grad.abs().max()
            assert float(e) < 1e-4, float(e)
    else:
        # tie-free planes agree with the single-workgroup form; in the plane with duplicates the winners may differ
        # between the two forms, the gradient mass of each duplicated pair may not
        (gf_s, gk_s, _), (gf_1, gk_1, _) = outs
        assert torch.equal(gf_s[1:], gf_1[1:]) and torch.equal(gf_s[0, C:], gf_1[0, C:])
        assert not torch.isnan(gf_s).any() and not torch.isnan(gk_s).any()
        pair = lambda t, lo: t[0, :lo, :64] + t[0, :lo, half:half + 64]
        assert float((pair(gf_s, C) - pair(gf_1, C)).abs().max()) <= 1e-5 * float(gf_1.abs().max())
        if B > 1:
            assert float(((gk_s - gk_1)[1:]).abs().max()) <= 2e-6 * float(gk_1.abs().max())
        assert float(((gk_s - gk_1)[0, dim:]).abs().max()) <= 2e-6 * float(gk_1.abs().max())
        assert float((pair(gk_s, dim) - pair(gk_1, dim)).abs().max()) <= 1e-4 * float(gk_1.abs().max())


@pytest.mark.parametrize("shape", [(2, 4096, 8, 16, 32, 2), (4, 4096, 16, 16, 16, 2), (2, 2048, 8, 8, 32, 2), (1, 4096, 4, 32, 16, 2)],
                         ids=lambda s: "B%dN%dH%dC%dW%dD%d" % s)
@pytest.mark.parametrize("dup", ["three_points", "half_the_cloud"])
def test_a_tied_channel_group_is_redone_alone(shape, dup):
    """Splat(max) backward with the plane's points in registers keeps its tie test per four-channel group and redoes only a tied
    group (ct_raster_hot.h: splat_bwd_quad<.., DELTA>: rows rewritten with single-winner claims, g_keys corrected by the
    difference).  Against the generic kernels on the same data: duplicated points tie bit for bit, so per point only the SUM
    over the copies is defined — it must agree; without duplicates everything must."""
    from cloud_transformers_amd import _lib
    from cloud_transformers_amd.step import SplatSliceStep
    lib = _lib.load()
    B, N, H, C, W, dim = shape
    g = torch.Generator(device="cuda").manual_seed(11)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda", generator=g))
    feat = torch.randn(B, H * C, N, device="cuda", generator=g)
    cot = torch.randn(B, H * C, N, device="cuda", generator=g)
    half = N // 2
    if dup == "half_the_cloud":
        src = torch.arange(half, device="cuda")
    else:
        src = torch.tensor([5, 77, half - 1], device="cuda")           # isolated ties: three planes' worth of single tied groups
    keys[..., half + src] = keys[..., src]
    feat[..., half + src] = feat[..., src]
    out = {}
    for name, flags in (("hot", _lib.DEBUG_FORCE_HOT), ("generic", _lib.DEBUG_NO_HOT)):
        lib.ct_debug_set_flags(flags)
        try:
            st = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=False)
            st.run()
            torch.cuda.synchronize()
            out[name] = (st.g_feat.clone(), st.g_keys_out.clone(), st.launch_tags()["splat_bwd"])
        finally:
            lib.ct_debug_set_flags(0)
    assert out["hot"][2].startswith("splat_max_bwd_hot") and "hot" not in out["generic"][2], (out["hot"][2], out["generic"][2])

    def folded(t):          # a duplicated point and its copy as one
        t = t.clone()
        t[..., src] += t[..., half + src]
        t[..., half + src] = 0
        return t
    for i, name in ((0, "g_feat"), (1, "g_keys")):
        a, b = folded(out["hot"][i]), folded(out["generic"][i])
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))

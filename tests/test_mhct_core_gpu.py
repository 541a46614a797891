"""Parity of the plane-resident MHCT core (ct_mhct_core_fwd / _bwd) — Splat -> grouped conv -> Slice
of a (batch, head) plane in one kernel — against the oracle (oracle/ref_cpu.py pieces + torch's CPU convolution) and
against this package's own unfused chain (ct_splat_fwd -> ct_gconv_fwd -> ct_slice_fwd) on identical inputs.

Bars: the rasterised grid z bit-exact (and the occupancy count exact); conv output y, sliced features and every
gradient within 1e-4 of the tensor's max (the north star's bar; observed ~1e-6).  Every cluster size (workgroups
per plane exchanging partial grids through the workspace) is forced on the small shapes; the full-size shapes run the
planner's own choice, several launches with fresh inputs (a stale exchange would show up as a wrong z), and the
workspace's status word must stay 0."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

SHAPES = [(2, 32, 16), (2, 16, 16), (3, 8, 32)]       # (dim, W, C): the three grids the kernel is built for


def _libs():
    from cloud_transformers_amd import _lib
    return _lib, _lib.load()


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def make_inputs(B, H, C, N, dim, seed, pad_kind=None, dup=False):
    g = torch.Generator().manual_seed(seed)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    if dup:                                   # duplicated points: exact ties in the max
        keys[..., N // 2:] = keys[..., :N - N // 2]
        feat[..., N // 2:] = feat[..., :N - N // 2]
    taps = 3 ** dim
    w = torch.randn(H * C, C, *([3] * dim), generator=g) / (C * taps) ** 0.5
    bias = torch.randn(H * C, generator=g) * 0.1
    cot = torch.randn(B, H * C, N, generator=g)
    pad = None
    if pad_kind == "f32":
        pad = (torch.rand(B, N, generator=g) > 0.25).float()
    elif pad_kind == "i32":
        pad = (torch.rand(B, N, generator=g) > 0.25).to(torch.int32)
    return keys, feat, w, bias, cot, pad


def oracle_core(keys, feat, w, bias, cot, pad, W, H, dim):
    k = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    bt = bias.clone().requires_grad_(True)
    lc, idx = R.positions(k, W, H, dim)
    z = R.splat(lc, idx, f, pad, W, H, dim, "max")
    conv = F.conv2d if dim == 2 else F.conv3d
    y = conv(z, wt, bt, padding=1, groups=H)
    out = R.slice_(lc, idx, y, pad, W, H, dim)
    out.backward(cot)
    occ = int((z.detach().abs() > 1e-9).sum())
    return dict(z=z.detach(), y=y.detach(), out=out.detach(), occ=occ, g_keys=k.grad, g_feat=f.grad, g_w=wt.grad, g_b=bt.grad)


def fused_core(keys, feat, w, bias, cot, pad, W, H, dim, want_grids=True, ws=None):
    """straight through the C ABI"""
    from cloud_transformers_amd.ops import _ptr, _stream, _pad_args
    L, lib = _libs()
    dev = "cuda"
    keys, feat, w, bias, cot = (t.to(dev).contiguous() for t in (keys, feat, w, bias, cot))
    B, HC, N = feat.shape
    C = HC // H
    Wl = [W] * dim
    Wa = L.int_array(Wl)
    padt, code = _pad_args(pad.to(dev) if pad is not None else None, B, N)
    out = torch.empty(B, HC, N, device=dev)
    z = torch.full((B, HC, *Wl), float("nan"), device=dev) if want_grids else None
    y = torch.full((B, HC, *Wl), float("nan"), device=dev) if want_grids else None
    occ = torch.full((), -1, device=dev, dtype=torch.int64)
    nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
    if ws is None:
        ws = torch.full((nws,), 0xAB, device=dev, dtype=torch.uint8)        # garbage: the init call must be all it takes
        L.check(lib.ct_mhct_core_workspace_init(_ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "ct_mhct_core_workspace_init")
    assert ws.numel() >= nws
    L.check(lib.ct_mhct_core_fwd(_ptr(keys), _ptr(feat), _ptr(padt), code, _ptr(w), _ptr(bias), _ptr(out), _ptr(z), _ptr(y),
                                 _ptr(occ), _ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "ct_mhct_core_fwd")
    st = ctypes.c_int(-1)
    L.check(lib.ct_mhct_core_status(_ptr(ws), nws, B, H, C, N, dim, Wa, ctypes.byref(st), _stream()), "ct_mhct_core_status")
    res = dict(z=z, y=y, out=out, occ=int(occ), status=st.value)
    if want_grids:
        g_feat, g_keys, g_w, g_b = torch.empty_like(feat), torch.empty_like(keys), torch.empty_like(w), torch.empty_like(bias)
        nb = lib.ct_mhct_core_bwd_workspace_bytes(B, H, C, N, dim, Wa)
        wsb = torch.empty(nb, device=dev, dtype=torch.uint8)
        L.check(lib.ct_mhct_core_bwd(_ptr(keys), _ptr(feat), _ptr(padt), code, _ptr(w), _ptr(z), _ptr(y), _ptr(cot), _ptr(g_feat),
                                     _ptr(g_keys), _ptr(g_w), _ptr(g_b), _ptr(wsb), nb, B, H, C, N, dim, Wa, _stream()),
                "ct_mhct_core_bwd")
        res.update(g_feat=g_feat, g_keys=g_keys, g_w=g_w, g_b=g_b)
    torch.cuda.synchronize()
    return res


def unfused_chain(keys, feat, w, bias, pad, W, H, dim):
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers.gconv import GroupedConvFn
    dev = "cuda"
    keys, feat, w, bias = (t.to(dev) for t in (keys, feat, w, bias))
    pad = pad.to(dev) if pad is not None else None
    z = ops.splat_keys(keys, feat, pad, W, H, dim, "max")
    y = GroupedConvFn.apply(z, w, bias, H)
    out = ops.slice_keys(keys, y, pad, W, H, dim)
    return dict(z=z, y=y, out=out, occ=int(ops.grid_occupancy_count(z)))


@pytest.fixture
def core_flags():
    _, lib = _libs()
    yield lib.ct_debug_set_core
    lib.ct_debug_set_core(0)


@pytest.mark.parametrize("dim,W,C", SHAPES)
@pytest.mark.parametrize("S", [1, 2, 4, 8])
@pytest.mark.parametrize("pad_kind", [None, "f32"])
def test_core_matches_oracle_every_cluster_size(core_flags, dim, W, C, S, pad_kind):
    core_flags(S << 8)
    B, H, N = 2, 4, 2048
    keys, feat, w, bias, cot, pad = make_inputs(B, H, C, N, dim, 100 + S, pad_kind)
    ref = oracle_core(keys, feat, w, bias, cot, pad, W, H, dim)
    got = fused_core(keys, feat, w, bias, cot, pad, W, H, dim)
    assert got["status"] == 0
    assert torch.equal(got["z"].cpu(), ref["z"]), "rasterised grid must be bit-exact"
    assert got["occ"] == ref["occ"]
    for name in ("y", "out", "g_keys", "g_feat", "g_w", "g_b"):
        assert relerr(got[name], ref[name]) <= 1e-4, (name, relerr(got[name], ref[name]))


@pytest.mark.parametrize("dim,W,C", SHAPES)
def test_core_ragged_tail_ties_and_int_padding(core_flags, dim, W, C):
    core_flags(0)
    B, H, N = 3, 5, 1020                      # planes not a multiple of 8, N not a multiple of the workgroup's quads
    keys, feat, w, bias, cot, pad = make_inputs(B, H, C, N, dim, 7, "i32", dup=True)
    ref = oracle_core(keys, feat, w, bias, cot, pad, W, H, dim)
    got = fused_core(keys, feat, w, bias, cot, pad, W, H, dim)
    assert got["status"] == 0
    assert torch.equal(got["z"].cpu(), ref["z"])
    assert got["occ"] == ref["occ"]
    for name in ("y", "out"):
        assert relerr(got[name], ref[name]) <= 1e-4, (name, relerr(got[name], ref[name]))


@pytest.mark.parametrize("dim,W,C", SHAPES)
def test_core_without_side_outputs(core_flags, dim, W, C):
    """inference form: z and y never leave the chip; `out` must not depend on the side outputs being requested"""
    core_flags(0)
    B, H, N = 2, 8, 4096
    keys, feat, w, bias, cot, pad = make_inputs(B, H, C, N, dim, 11)
    a = fused_core(keys, feat, w, bias, cot, pad, W, H, dim, want_grids=True)
    b = fused_core(keys, feat, w, bias, cot, pad, W, H, dim, want_grids=False)
    assert torch.equal(a["out"], b["out"]) and a["occ"] == b["occ"] and b["status"] == 0


FULL = [  # (B, H, N, dim, W, C): the headline shape and the stage-3 zoo heads at training and decoder sizes
    (8, 64, 4096, 2, 32, 16),
    (8, 16, 4096, 2, 16, 16),
    (8, 16, 4096, 3, 8, 32),
    (2, 16, 16384, 2, 16, 16),
    (2, 16, 16384, 3, 8, 32),
]


@pytest.mark.parametrize("B,H,N,dim,W,C", FULL)
def test_core_full_size_vs_unfused_chain_and_oracle_planes(core_flags, B, H, N, dim, W, C):
    core_flags(0)
    for rep in range(3):                      # fresh inputs per launch: a stale exchange buffer would give a wrong z
        keys, feat, w, bias, cot, pad = make_inputs(B, H, C, N, dim, 1000 + rep)
        got = fused_core(keys, feat, w, bias, cot, pad, W, H, dim)
        ref = unfused_chain(keys, feat, w, bias, pad, W, H, dim)
        assert got["status"] == 0
        assert torch.equal(got["z"], ref["z"]), "z differs from ct_splat_fwd (rep %d)" % rep
        assert got["occ"] == ref["occ"]
        assert relerr(got["y"], ref["y"]) <= 1e-5
        assert relerr(got["out"], ref["out"]) <= 1e-5
    # sampled planes against the oracle (planes are independent), gradients included
    rng = np.random.RandomState(0)
    for _ in range(3):
        b, h = int(rng.randint(B)), int(rng.randint(H))
        sl = lambda t, per: t[b:b + 1, h * per:(h + 1) * per]
        kp, fp, cp = sl(keys, dim), sl(feat, C), sl(cot, C)
        wp, bp = w[h * C:(h + 1) * C], bias[h * C:(h + 1) * C]
        ref1 = oracle_core(kp, fp, wp, bp, cp, None, W, 1, dim)
        assert torch.equal(sl(got["z"], C).cpu(), ref1["z"])
        for name, per in (("out", C), ("g_feat", C), ("g_keys", dim)):
            assert relerr(sl(got[name], per), ref1[name]) <= 1e-4, (name, relerr(sl(got[name], per), ref1[name]))


@pytest.mark.parametrize("B,H,N,dim,W,C", [(8, 16, 4096, 2, 16, 16), (2, 16, 16384, 3, 8, 32)])
def test_core_workspace_is_reusable_launch_after_launch(core_flags, B, H, N, dim, W, C):
    """ONE workspace, initialised once, for a train of launches on changing inputs: the clusters' counters must be back
    at zero after every launch and no launch may read a partner tile of the previous one."""
    from cloud_transformers_amd.ops import _ptr, _stream
    L, lib = _libs()
    core_flags(0)
    Wa = L.int_array([W] * dim)
    nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
    ws = torch.full((nws,), 0x5C, device="cuda", dtype=torch.uint8)
    L.check(lib.ct_mhct_core_workspace_init(_ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "init")
    sets = [make_inputs(B, H, C, N, dim, 40 + i) for i in range(2)]
    refs = [unfused_chain(k, f, w, b, None, W, H, dim) for k, f, w, b, _, _ in sets]
    for rep in range(12):
        keys, feat, w, bias, cot, _ = sets[rep % 2]
        got = fused_core(keys, feat, w, bias, cot, None, W, H, dim, want_grids=(rep % 3 != 2), ws=ws)
        assert got["status"] == 0
        if got["z"] is not None:
            assert torch.equal(got["z"], refs[rep % 2]["z"]), rep
        assert got["occ"] == refs[rep % 2]["occ"]
        assert relerr(got["out"], refs[rep % 2]["out"]) <= 1e-5


@pytest.mark.parametrize("B,H,N,dim,W,C", [(8, 16, 4096, 3, 8, 32), (2, 16, 16384, 2, 16, 16)])
def test_core_clusters_under_uneven_load(core_flags, B, H, N, dim, W, C):
    """The cluster hand-off (partial grids through the workspace, agent-scope arrive / acquire) with the chip busy and unevenly
    loaded: a stream of large copies runs beside 40 back-to-back launches on alternating inputs; every launch's z must be the
    three-kernel chain's bit for bit (idle chips and uniform load hide stale reads: MI355X_MICROARCH.md, hand-off testing)."""
    from cloud_transformers_amd.ops import _ptr, _stream
    L, lib = _libs()
    core_flags(0)
    Wl = [W] * dim
    Wa = L.int_array(Wl)
    sets = []
    for i in range(2):
        keys, feat, w, bias, _, _ = make_inputs(B, H, C, N, dim, 300 + i)
        ref = unfused_chain(keys, feat, w, bias, None, W, H, dim)
        sets.append(([t.cuda().contiguous() for t in (keys, feat, w, bias)], ref["z"].clone(), ref["occ"]))
    nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
    ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
    L.check(lib.ct_mhct_core_workspace_init(_ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "init")
    out = torch.empty(B, H * C, N, device="cuda")
    zs = [torch.empty(B, H * C, *Wl, device="cuda") for _ in range(40)]
    y = torch.empty(B, H * C, *Wl, device="cuda")
    occs = torch.zeros(40, device="cuda", dtype=torch.int64)
    side = torch.cuda.Stream()
    big_a = torch.empty(96 << 20, device="cuda", dtype=torch.uint8)         # a few CUs' worth of copy kernels at a time
    big_b = torch.empty_like(big_a)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(60):
            big_b[: (17 << 20)].copy_(big_a[: (17 << 20)])
            big_a.copy_(big_b)
    for it in range(40):
        (keys, feat, w, bias), _, _ = sets[it % 2]
        L.check(lib.ct_mhct_core_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(bias), _ptr(out), _ptr(zs[it]), _ptr(y),
                                     occs[it:].data_ptr(), _ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "core")
    torch.cuda.synchronize()
    st = ctypes.c_int(-1)
    L.check(lib.ct_mhct_core_status(_ptr(ws), nws, B, H, C, N, dim, Wa, ctypes.byref(st), _stream()), "status")
    assert st.value == 0
    for it in range(40):
        assert torch.equal(zs[it], sets[it % 2][1]), "launch %d read a stale or partial grid" % it
        assert int(occs[it]) == sets[it % 2][2]


def fused_bwd(keys, feat, w, bias, cot, pad, W, H, dim):
    from cloud_transformers_amd.ops import _ptr, _stream, _pad_args
    L, lib = _libs()
    keys, feat, w, bias, cot = (t.cuda().contiguous() for t in (keys, feat, w, bias, cot))
    B, HC, N = feat.shape
    C = HC // H
    Wa = L.int_array([W] * dim)
    padt, code = _pad_args(pad.cuda() if pad is not None else None, B, N)
    g_feat, g_keys, g_w, g_b = (torch.full_like(t, float("nan")) for t in (feat, keys, w, bias))
    nws = lib.ct_mhct_core_bwd_fused_workspace_bytes(B, H, C, N, dim, Wa)
    ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
    L.check(lib.ct_mhct_core_bwd_fused(_ptr(keys), _ptr(feat), _ptr(padt), code, _ptr(w), _ptr(bias), _ptr(cot), _ptr(g_feat),
                                       _ptr(g_keys), _ptr(g_w), _ptr(g_b), _ptr(ws), nws, B, H, C, N, dim, Wa, _stream()),
            "ct_mhct_core_bwd_fused")
    torch.cuda.synchronize()
    return dict(g_feat=g_feat, g_keys=g_keys, g_w=g_w, g_b=g_b)


@pytest.mark.parametrize("B,H,N,pad_kind", [(2, 4, 2048, None), (3, 5, 1020, "f32"), (2, 3, 5000, "i32"), (1, 2, 16384, None)])
def test_lds_resident_backward_matches_oracle(B, H, N, pad_kind):
    """ct_mhct_core_bwd_fused (2D 16^2 C16: z and conv(z) recomputed in LDS, nothing saved by the forward) against the oracle's
    autograd through positions -> splat -> conv -> slice: every cotangent within 1e-4 of its maximum."""
    dim, W, C = 2, 16, 16
    keys, feat, w, bias, cot, pad = make_inputs(B, H, C, N, dim, 900 + N, pad_kind)
    ref = oracle_core(keys, feat, w, bias, cot, pad, W, H, dim)
    got = fused_bwd(keys, feat, w, bias, cot, pad, W, H, dim)
    for name in ("g_keys", "g_feat", "g_w", "g_b"):
        assert torch.isfinite(got[name]).all(), name
        assert relerr(got[name], ref[name]) <= 1e-4, (name, relerr(got[name], ref[name]))


def test_lds_resident_backward_full_size_and_reproducible():
    """B8 H16 N4096 (the stage-3 zoo head): against ct_mhct_core_bwd on saved grids, twice (bitwise equal: fixed-point
    accumulation, fixed-order sums over the batch), and exact ties route to a single winner."""
    dim, W, C, B, H, N = 2, 16, 16, 8, 16, 4096
    keys, feat, w, bias, cot, _ = make_inputs(B, H, C, N, dim, 31)
    saved = fused_core(keys, feat, w, bias, cot, None, W, H, dim)
    a = fused_bwd(keys, feat, w, bias, cot, None, W, H, dim)
    b = fused_bwd(keys, feat, w, bias, cot, None, W, H, dim)
    for name in ("g_keys", "g_feat", "g_w", "g_b"):
        assert torch.equal(a[name], b[name]), name
        assert relerr(a[name], saved[name]) <= 1e-5, (name, relerr(a[name], saved[name]))
    # duplicated points: every tied pair sends the cell's cotangent to exactly one of the two
    keys, feat, w, bias, cot, _ = make_inputs(2, 4, C, 2048, dim, 32, dup=True)
    got = fused_bwd(keys, feat, w, bias, cot, None, W, 4, dim)
    gf = got["g_feat"].cpu()
    half = 1024
    both = (gf[..., :half] != 0) & (gf[..., half:] != 0)
    assert int(both.sum()) == 0
    ref = oracle_core(keys, feat, w, bias, cot, None, W, 4, dim)
    assert relerr(gf[..., :half] + gf[..., half:], ref["g_feat"][..., :half] + ref["g_feat"][..., half:]) <= 1e-4


def test_core_autograd_function_matches_module_chain(core_flags):
    """ops.mhct_core (autograd.Function over the ABI pair) against the unfused autograd chain of this package"""
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers.gconv import GroupedConvFn
    core_flags(0)
    for dim, W, C in SHAPES:
        B, H, N = 2, 16, 2048
        keys, feat, w, bias, cot, _ = make_inputs(B, H, C, N, dim, 5)
        outs = []
        for fused in (True, False):
            k, f, wt, bt = (t.cuda().requires_grad_(True) for t in (keys, feat, w, bias))
            if fused:
                out, occ = ops.mhct_core(k, f, None, wt, bt, W, H, dim)
            else:
                z = ops.splat_keys(k, f, None, W, H, dim, "max")
                out = ops.slice_keys(k, GroupedConvFn.apply(z, wt, bt, H), None, W, H, dim)
                occ = ops.grid_occupancy_count(z)
            out.backward(cot.cuda())
            outs.append((out.detach(), int(occ), k.grad, f.grad, wt.grad, bt.grad))
        assert outs[0][1] == outs[1][1]
        for a, b in zip(outs[0], outs[1]):
            if torch.is_tensor(a):
                assert relerr(a, b) <= 1e-5


@pytest.mark.parametrize("shape", [(8, 16, 4096, 32, 8, 3), (8, 16, 4096, 16, 16, 2)], ids=["8^3C32", "16^2C16"])
def test_a_cluster_timeout_is_loud_and_the_workspace_recovers(shape):
    """a workgroup that gives up waiting for its partners used to return, leaving `out` unwritten and the plane's
    counters non-zero for every later launch on the cached workspace.  Fault injection (ct_debug_set_core bit 1: the last
    workgroup of plane 0 arrives late, its partners time out after a short spin): the status word is set, the timed-out
    workgroups' share of the outputs is NaN (nothing else is), the counters are zero again — the NEXT launch on the same
    workspace is correct without re-initialising — and ops.mhct_core_check() raises and re-initialises."""
    from cloud_transformers_amd import ops
    L, lib = _libs()
    B, H, N, C, W, dim = shape
    g = torch.Generator().manual_seed(3)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    w = torch.randn(H * C, C, *([3] * dim), generator=g) / (C * 3 ** dim) ** 0.5
    bias = torch.randn(H * C, generator=g) * 0.1
    cot = torch.randn(B, H * C, N, generator=g)
    force = 2 << 8                    # clusters of two workgroups per plane (the light 16^2 plane runs unclustered by itself)
    lib.ct_debug_set_core(force)
    try:
        good = fused_core(keys, feat, w, bias, cot, None, W, H, dim, want_grids=False)
        assert good["status"] == 0
        Wa = L.int_array([W] * dim)
        nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
        ws = torch.zeros(nws, device="cuda", dtype=torch.uint8)
        L.check(lib.ct_mhct_core_workspace_init(ops._ptr(ws), nws, B, H, C, N, dim, Wa, ops._stream()), "init")
        lib.ct_debug_set_core(force | 2)
        hurt = fused_core(keys, feat, w, bias, cot, None, W, H, dim, want_grids=False, ws=ws)
        lib.ct_debug_set_core(force)
        again = fused_core(keys, feat, w, bias, cot, None, W, H, dim, want_grids=False, ws=ws)
    finally:
        lib.ct_debug_set_core(0)
    assert hurt["status"] == 1
    nan = torch.isnan(hurt["out"])
    assert nan.any() and not nan[1:].any() and not nan[0, C:].any()          # plane (0, 0) only
    ok = ~nan
    assert torch.equal(hurt["out"][ok], good["out"][ok])
    # the same workspace, not re-initialised: counters are back at zero, the launch is right (the status word stays set until read)
    assert torch.equal(again["out"], good["out"]) and again["occ"] == good["occ"]
    # the product's check: raises once, re-initialises, then is quiet
    key = (0, torch.cuda.current_stream().cuda_stream, B, H, C, N, tuple([W] * dim))
    saved = ops._core_ws.get(key)
    ops._core_ws[key] = ws
    try:
        with pytest.raises(RuntimeError, match="timed out"):
            ops.mhct_core_check()
        assert ops.mhct_core_check() == []
    finally:
        if saved is None:
            del ops._core_ws[key]
        else:
            ops._core_ws[key] = saved

"""Arrival tickets (include/cloudct.h: ct_slice_bwd_tk / ct_splat_bwd_tk): the sums over the workgroups that share a (b,h)
plane — partial g_keys of the channel-chunk groups, partial g_grid tiles of the point segments — happen inside the
backward kernels instead of in sum_parts launches behind them.

The fold adds the partials in the order of the two-launch form, so the two forms must agree BIT FOR BIT on every
cotangent; the two-launch form itself is held to the oracle by test_raster_gpu.py / test_headline_gpu.py.  A hand-off
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
        tk.run()
        two.run()
        torch.cuda.synchronize()
        for name in ("z", "out", "g_z", "g_feat", "g_keys_buf"):
            a, b = getattr(tk, name), getattr(two, name)
            assert torch.equal(a, b), (name, it, float((a - b).abs().max()))
        assert int(tk.tickets.abs().sum()) == 0, "tickets not reset"
        tags = tk.launch_tags()
        torch.cuda.synchronize()
        folded = folded or any("folded" in t for t in tags.values())
        assert int(tk.tickets.abs().sum()) == 0
    B, N, H, C, W, dim = shape
    if B * H < 256 and C >= 16:            # every such shape shares planes between workgroups in at least one pass
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
            for name in ("g_z", "g_feat", "g_keys_buf"):
                assert torch.equal(getattr(t, name), getattr(r, name)), (name, it)
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
    assert "folded" in tag1 and "folded" not in tag0, (tag1, tag0)
    assert torch.equal(gk1, gk0) and torch.equal(gf1, gf0)

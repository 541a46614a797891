"""API behaviour around the kernels: non-contiguous inputs, partial requires_grad, no_grad,
dtype checks — the things a model_zoo file can throw at the modules."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _inputs(B=2, H=3, C=4, N=200, dim=2, W=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
    feat = torch.randn(B, H * C, N, generator=g)
    return keys, feat


def test_non_contiguous_inputs_match_contiguous():
    from cloud_transformers_amd import ops
    keys, feat = _inputs()
    kc, fc = keys.cuda(), feat.cuda()
    # channel-sliced views (what `key_values[:, :3H]` / `[:, 3H:]` produce) and a transposed layout
    big = torch.cat([torch.zeros_like(fc), fc], dim=1)[:, fc.shape[1]:]
    tr = fc.transpose(1, 2).contiguous().transpose(1, 2)
    assert not big.is_contiguous() and not tr.is_contiguous()
    z0 = ops.splat_keys(kc, fc, None, [8, 8], 3, 2)
    assert torch.equal(z0, ops.splat_keys(kc, big, None, [8, 8], 3, 2))
    assert torch.equal(z0, ops.splat_keys(kc, tr, None, [8, 8], 3, 2))
    o0 = ops.slice_keys(kc, z0, None, [8, 8], 3, 2)
    zv = torch.stack([z0, z0], 0)[1]
    assert torch.equal(o0, ops.slice_keys(kc, zv, None, [8, 8], 3, 2))


def test_partial_requires_grad_and_no_grad():
    from cloud_transformers_amd import ops
    keys, feat = _inputs()
    kc = keys.cuda()
    fc = feat.cuda().requires_grad_(True)
    z = ops.splat_keys(kc, fc, None, [8, 8], 3, 2)        # keys carry no gradient
    o = ops.slice_keys(kc, z, None, [8, 8], 3, 2)
    o.sum().backward()
    assert fc.grad is not None and torch.isfinite(fc.grad).all()
    with torch.no_grad():
        z2 = ops.splat_keys(kc, fc, None, [8, 8], 3, 2)
    assert not z2.requires_grad and torch.equal(z2, z.detach())


def test_dtype_and_shape_errors():
    from cloud_transformers_amd.layers.cloud_transform import DifferentiablePositions, Splat
    keys, feat = _inputs()
    with pytest.raises(AssertionError):
        Splat(8, 3, 2).cuda().forward_keys(keys.cuda(), feat.cuda().double())       # reference asserts float32
    with pytest.raises(AssertionError):
        DifferentiablePositions(8, 3, 2).cuda()(keys.cuda()[:, :5])                  # keys.size(1) != heads*dim
    with pytest.raises(TypeError):
        from cloud_transformers_amd import ops
        lc, idx = ops.positions(keys.cuda(), 8, 3, 2)
        ops.splat_lc(lc, idx.int(), feat.cuda(), None, 8, 3, 2)                      # flattened_index must be int64


def test_int_and_bool_like_padding_masks():
    from cloud_transformers_amd import ops
    keys, feat = _inputs()
    g = torch.Generator().manual_seed(3)
    pad = (torch.rand(2, 200, generator=g) > 0.3)
    lc, idx = R.positions(keys, 8, 3, 2)
    ref = R.splat(lc, idx, feat, pad.float(), 8, 3, 2)
    for p in (pad.float(), pad.int(), pad.long(), pad):
        z = ops.splat_keys(keys.cuda(), feat.cuda(), p.cuda(), [8, 8], 3, 2)
        assert torch.equal(z.cpu(), ref), p.dtype


def test_modules_move_with_to_and_eval():
    from cloud_transformers_amd.layers.multihead_ct import MultiHead
    m = MultiHead(16, 4, 16, 8, 2, 2)
    m = m.to("cuda").eval()
    x = torch.randn(1, 16, 64, device="cuda")
    pcd = torch.rand(1, 3, 64, device="cuda") * 2 - 1
    with torch.no_grad():
        (res, lattice), stats = m(x, pcd, return_lattice=True)
    assert res.shape == (1, 8, 64) and lattice.shape == (1, 4, 64) and len(stats) == 4 and stats[3] is None
    assert m.splat.tensor_mod.device.type == "cuda"



def test_empty_clouds_and_batches():
    """Empty inputs: an empty cloud rasterises to the zero floor and slices to an empty tensor; gradients
    are zeros of the right shapes."""
    from cloud_transformers_amd import ops
    dev = "cuda"
    keys = torch.zeros(2, 3 * 2, 0, device=dev, requires_grad=True)
    feat = torch.zeros(2, 3 * 4, 0, device=dev, requires_grad=True)
    z = ops.splat_keys(keys, feat, None, [8, 8], 3, 2)
    assert z.shape == (2, 12, 8, 8) and float(z.abs().max()) == 0.0
    grid = torch.randn(2, 12, 8, 8, device=dev, requires_grad=True)
    o = ops.slice_keys(keys, grid, None, [8, 8], 3, 2)
    assert o.shape == (2, 12, 0)
    (z.sum() + o.sum()).backward()
    assert feat.grad.shape == feat.shape and keys.grad.shape == keys.shape and float(grid.grad.abs().max()) == 0.0
    lc, idx = ops.positions(keys, 8, 3, 2)
    assert lc.shape == (2, 3, 4, 0) and idx.dtype == torch.int64
    zb = ops.splat_keys(torch.zeros(0, 6, 5, device=dev), torch.zeros(0, 12, 5, device=dev), None, [8, 8], 3, 2)
    assert zb.shape == (0, 12, 8, 8)

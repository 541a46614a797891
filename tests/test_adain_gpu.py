"""Fused adaptive instance norm (ct_adain_fwd / ct_adain_bwd) against the CPU oracle's
restatement of AdaIn1dUpd (oracle/ref_cpu.py:_adain, reference layers/utils.py:82-97),
forward and backward (oracle gradients by torch autograd in float64), tolerance 1e-4."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _oracle(x, style, w, b, relu, gy):
    x = x.double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    b = b.double().requires_grad_(True)
    y = R._adain(x, style.double(), {"p.linear.weight": w, "p.linear.bias": b}, "p")
    if relu:
        y = torch.relu(y)
    y.backward(gy.double())
    return y.detach(), x.grad, w.grad, b.grad


# register path (N % 4 == 0, <= 16384) at every NV, strided path (ragged N, long rows), tiny rows
@pytest.mark.parametrize("B,C,N", [(2, 48, 1024), (2, 16, 2048), (1, 12, 4096), (2, 6, 8192), (1, 5, 16384),
                                   (2, 7, 1001), (1, 3, 20000), (3, 4, 4), (2, 5, 1), (1, 2, 300)])
@pytest.mark.parametrize("relu", [False, True])
def test_adain_matches_oracle(B, C, N, relu):
    from cloud_transformers_amd.layers.utils import AdaIn1dUpd
    g = torch.Generator().manual_seed(B * 7919 + C * 31 + N)
    L = 16
    mod = AdaIn1dUpd(C, L)
    with torch.no_grad():
        mod.linear.weight.copy_(torch.randn(2 * C, L, generator=g) * 0.5)
        mod.linear.bias.copy_(torch.randn(2 * C, generator=g) * 0.5)
    x = torch.randn(B, C, N, generator=g) * 3 + torch.randn(B, C, 1, generator=g) * 5     # rows far from zero mean
    style = torch.randn(B, L, generator=g)
    gy = torch.randn(B, C, N, generator=g)
    yo, gxo, gwo, gbo = _oracle(x, style, mod.linear.weight.detach(), mod.linear.bias.detach(), relu, gy)

    mod = mod.cuda()
    xc = x.cuda().requires_grad_(True)
    y = mod(xc, style.cuda(), relu=relu)
    y.backward(gy.cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.numpy(), atol=TOL, rtol=TOL)
    if N > 1:      # a single-point row has zero variance: xhat = 0 and d/dx is ~0 * rsqrt(eps), ill-conditioned in fp32
        scale = max(1.0, float(gxo.abs().max()))
        np.testing.assert_allclose(xc.grad.cpu().numpy(), gxo.numpy(), atol=TOL * scale, rtol=1e-3)
    scale = max(1.0, float(gwo.abs().max()))
    np.testing.assert_allclose(mod.linear.weight.grad.cpu().numpy(), gwo.numpy(), atol=TOL * scale, rtol=1e-3)
    np.testing.assert_allclose(mod.linear.bias.grad.cpu().numpy(), gbo.numpy(), atol=TOL * scale, rtol=1e-3)


def test_adain_is_deterministic_and_leaves_inputs_alone():
    from cloud_transformers_amd import ops
    x = torch.randn(2, 32, 4096, device="cuda")
    gb = torch.randn(2, 2, 32, device="cuda")
    x0, gb0 = x.clone(), gb.clone()
    a = ops.adain(x, gb, 1e-5, True)
    b = ops.adain(x, gb, 1e-5, True)
    assert torch.equal(a, b) and torch.equal(x, x0) and torch.equal(gb, gb0)
    assert float(a.min()) >= 0.0


def test_forward_style_fuses_the_following_relu():
    """`after` = Sequential(AdaIn1dUpd, ReLU(inplace)) (multihead_ct_adain.py:64-66): same result either way,
    and the layout of the Sequential (state-dict keys) is untouched."""
    from cloud_transformers_amd.layers.multihead_ct import forward_style
    from cloud_transformers_amd.layers.utils import AdaIn1dUpd
    torch.manual_seed(3)
    seq = torch.nn.Sequential(AdaIn1dUpd(24, 8), torch.nn.ReLU(inplace=True)).cuda()
    assert sorted(seq.state_dict()) == ["0.linear.bias", "0.linear.weight"]
    x = torch.randn(2, 24, 2048, device="cuda")
    z = torch.randn(2, 8, device="cuda")
    fused = forward_style(seq, x, z)
    plain = torch.relu(seq[0](x, z))
    assert torch.equal(fused, plain)
    # non-fp32 / non-3D inputs take torch's composition and still agree
    ref = torch.relu(seq[0].double()(x.double(), z.double()))
    np.testing.assert_allclose(fused.detach().cpu().numpy(), ref.detach().float().cpu().numpy(), atol=TOL)


def test_adain_abi_rejects_bad_arguments():
    from cloud_transformers_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 1, 4, device="cuda")
    assert lib.ct_adain_fwd(None, 0, None, None, 0, None, 0, None, None, 1, 1, 4, 1e-5, 0, None) == -1
    p = x.data_ptr()
    assert lib.ct_adain_fwd(p, 0, p, None, 0, p, 0, p, p, -1, 1, 4, 1e-5, 0, None) == -1
    assert lib.ct_adain_fwd(p, 2, p, None, 0, p, 0, p, p, 1, 1, 4, 1e-5, 0, None) == -1      # batch stride < C*N
    assert lib.ct_adain_fwd(None, 0, None, None, 0, None, 0, None, None, 0, 8, 4, 1e-5, 0, None) == 0      # empty batch: nothing to do


@pytest.mark.parametrize("N", [4096, 20000])          # register-resident rows and the strided kernel
def test_relu_mask_in_backward_is_the_forward_mask(N):
    """The backward recomputes the ReLU mask instead of reading the output: it must be the forward's mask bit for bit
    (an element within rounding of zero decided differently would move a whole cotangent) — g_beta[b,c] = sum_n gy * [y > 0]."""
    from cloud_transformers_amd import ops
    torch.manual_seed(N)
    B, C = 3, 68
    x = (torch.randn(B, C, N, device="cuda") * 4 + 2.5).requires_grad_(True)
    gb = torch.randn(B, 2, C, device="cuda", requires_grad=True)
    gy = torch.randn(B, C, N, device="cuda")
    y = ops.adain(x, gb, 1e-5, True)
    y.backward(gy)
    expect = (gy.double() * (y.detach() > 0)).sum(dim=2)
    got = gb.grad[:, 1].double()
    assert float((got - expect).abs().max()) <= 2e-3, float((got - expect).abs().max())     # one wrong element would show as ~1


@pytest.mark.parametrize("N", [512, 250])            # float4 rows and the strided kernel
def test_channel_slices_residual_and_sliced_cotangent(N):
    """x as a channel slice of a wider tensor, a skip connection added in the same pass, and a cotangent that is a
    slice of a concatenation's: all read where they lie, same results as torch's composition."""
    from cloud_transformers_amd import ops
    torch.manual_seed(N)
    B, C = 3, 20
    wide = torch.randn(B, C + 12, N, device="cuda", requires_grad=True)
    gb = torch.randn(B, 2, C, device="cuda", requires_grad=True)
    res = torch.randn(B, C, N, device="cuda", requires_grad=True)
    cot = torch.randn(B, C + 4, N, device="cuda")

    def run(fused):
        for t in (wide, gb, res):
            t.grad = None
        x = wide[:, 12:]                                   # rows contiguous, batch stride (C + 12) * N
        if fused:
            y = ops.adain(x, gb, 1e-5, True, res)
        else:
            xh = torch.nn.functional.instance_norm(x.contiguous(), eps=1e-5)
            y = torch.relu(xh * (gb[:, 0, :, None] + 1) + gb[:, 1, :, None]) + res
        z = torch.cat([y, wide[:, :4] * 0], dim=1)
        (z * cot).sum().backward()
        return y.detach(), wide.grad.clone(), gb.grad.clone(), res.grad.clone()

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.allclose(u, v, rtol=1e-4, atol=2e-4), float((u - v).abs().max())

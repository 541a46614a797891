"""GPU parity of the fused lattice kernels (rigid transform of xyz + residual, tanh) against
the CPU oracle's rigid_transform (pinned on the reference's Vol/PlaneTransformer outputs) and
its autograd, for every parameter cotangent."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_so3", [False, True], ids=["so3-apart", "so3-inside"])
@pytest.mark.parametrize("dim,use_scales,use_kscale", [(2, False, False), (3, True, False), (2, True, True), (3, False, True)])
def test_lattice_fwd_bwd(dim, use_scales, use_kscale, fused_so3):
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.layers.utils import so3_exponential_map
    g = torch.Generator().manual_seed(10 * dim + use_scales + 2 * use_kscale)
    B, H, N = 2, 5, 333
    xyz = torch.rand(B, 3, N, generator=g) * 2 - 1
    res = torch.randn(B, H * 3, N, generator=g) * 0.3
    log_R = torch.randn(H, 3, generator=g)
    shift = torch.randn(H, 3, generator=g) * 0.1
    scales = (1 + 0.2 * torch.randn(H, dim, generator=g)) if use_scales else None
    kscale = torch.tensor(0.7) if use_kscale else None
    cot_l = torch.randn(B, H * dim, N, generator=g)
    cot_k = torch.randn(B, H * dim, N, generator=g) * 0.1

    leaves = [t.clone().requires_grad_(True) for t in (xyz, res, log_R, shift)]
    sc = scales.clone().requires_grad_(True) if use_scales else None
    ks = kscale.clone().requires_grad_(True) if use_kscale else None
    p = leaves[0][:, None] + (leaves[1] if ks is None else ks * leaves[1]).reshape(B, H, 3, N)
    keys_ref = R.rigid_transform(p, leaves[2], torch.zeros(H, 3) + leaves[3], sc, dim).reshape(B, H * dim, N)
    lat_ref = torch.tanh(keys_ref)
    ((lat_ref * cot_l).sum() + (keys_ref * cot_k).sum()).backward()

    dl = [t.clone().cuda().requires_grad_(True) for t in (xyz, res, log_R, shift)]
    dsc = scales.clone().cuda().requires_grad_(True) if use_scales else None
    dks = kscale.clone().cuda().requires_grad_(True) if use_kscale else None
    if fused_so3:        # ops.LatticeSo3Fn: the so3 map inside the launches, key statistics by the launch's last workgroup
        keys, lat, stats = ops.lattice_so3(dl[0], dl[1], dl[2], dl[3], dsc, dks, dim, with_stats=True)
        assert abs(float(stats[0]) - float(keys_ref.mean())) <= 1e-5 * max(1.0, abs(float(keys_ref.mean())))
        assert abs(float(stats[1]) - float(keys_ref.var())) <= 1e-4 * float(keys_ref.var())
    else:
        keys, lat = ops.lattice(dl[0], dl[1], so3_exponential_map(dl[2]), dl[3], dsc, dks, dim)
    ((lat * cot_l.cuda()).sum() + (keys * cot_k.cuda()).sum()).backward()

    def close(a, b, name, tol=2e-5):
        err = float((a.detach().cpu() - b.detach()).abs().max())
        assert err <= tol * max(1.0, float(b.abs().max())), (name, err)

    close(keys, keys_ref, "keys")
    close(lat, lat_ref, "lattice")
    for a, b, name in zip(dl, leaves, ("xyz", "residual", "log_R", "shift")):
        close(a.grad, b.grad, "g_" + name, 1e-4)
    if use_scales:
        close(dsc.grad, sc.grad, "g_scales", 1e-4)
    if use_kscale:
        close(dks.grad, ks.grad, "g_kscale", 1e-4)


def test_so3_exp_map_matches_oracle():
    """ct_so3_exp_fwd / _bwd against the oracle's Rodrigues map (pinned on the transforms golden) and its float64
    autograd, including rotations below the clamp (|v|^2 < eps: a and b stop depending on v) and a zero vector."""
    from cloud_transformers_amd.layers.utils import so3_exponential_map
    g = torch.Generator().manual_seed(5)
    v = torch.cat([torch.randn(40, 3, generator=g), torch.randn(8, 3, generator=g) * 3, torch.randn(8, 3, generator=g) * 1e-3,
                   torch.zeros(1, 3), torch.tensor([[0.02, 0.0, 0.0], [0.0, 0.005, 0.0]])])
    cot = torch.randn(v.shape[0], 3, 3, generator=g)
    vr = v.double().requires_grad_(True)
    Rr = R.so3_exp(vr)
    (Rr * cot.double()).sum().backward()
    vc = v.cuda().requires_grad_(True)
    Rc = so3_exponential_map(vc)
    (Rc * cot.cuda()).sum().backward()
    assert Rc.shape == (v.shape[0], 3, 3)
    assert float((Rc.detach().cpu().double() - Rr.detach()).abs().max()) <= 2e-6
    assert float((vc.grad.cpu().double() - vr.grad).abs().max()) <= 2e-5 * max(1.0, float(vr.grad.abs().max()))
    # rotations: R R^T = I
    eye = torch.eye(3, device="cuda").expand_as(Rc)
    assert float((Rc.detach() @ Rc.detach().transpose(1, 2) - eye).abs().max()) <= 1e-5


def test_fused_lattice_statistics_over_repeated_launches():
    """The ticket word of ct_lattice_so3_fwd resets itself: many launches on one stream, fresh inputs each, every launch's key
    statistics equal the keys' own (a stale ticket or a partial read too early would show)."""
    from cloud_transformers_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, N, dim = 8, 16, 4096, 3
    for it in range(12):
        xyz = torch.rand(B, 3, N, device="cuda", generator=g) * 2 - 1
        res = torch.randn(B, H * 3, N, device="cuda", generator=g) * (0.1 + 0.05 * it)
        log_R = torch.randn(H, 3, device="cuda", generator=g)
        shift = torch.randn(H, 3, device="cuda", generator=g) * 0.1
        keys, lat, stats = ops.lattice_so3(xyz, res, log_R, shift, None, None, dim, with_stats=True)
        kd = keys.double()
        assert abs(float(stats[0]) - float(kd.mean())) <= 1e-5
        assert abs(float(stats[1]) - float(kd.var())) <= 1e-4 * float(kd.var()), it

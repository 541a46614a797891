import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_headline_gpu import oracle_chain, relerr, NAMES
from cloud_transformers_amd.step import SplatSliceStep
C, Wn, dim, B, N = 4, 128, 2, 2, 16384
H, W = 16, [Wn] * dim
torch.manual_seed(77 + C + Wn)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
step = SplatSliceStep(keys, feat, cot, Wn, H, dim, "max")
step.run(); torch.cuda.synchronize()
print(step.launch_tags())
step.run(); torch.cuda.synchronize()
got = (step.z, step.out, step.g_z, step.g_feat, step.g_keys())
for (b, h) in ((0, 0), (B - 1, H - 1), (B // 2, 5)):
    ref = oracle_chain(keys[b:b + 1, h * dim:(h + 1) * dim].cpu(), feat[b:b + 1, h * C:(h + 1) * C].cpu(), cot[b:b + 1, h * C:(h + 1) * C].cpu(), W, 1, dim, "max")
    sl = slice(h * C, (h + 1) * C)
    mine = (got[0][b:b + 1, sl], got[1][b:b + 1, sl], got[2][b:b + 1, sl], got[3][b:b + 1, sl], got[4][b:b + 1, h * dim:(h + 1) * dim])
    for name, a, r in zip(NAMES, mine, ref):
        d = (a.cpu() - r).abs()
        bad = (d > 1e-4 * r.abs().max()).nonzero()
        print((b, h), name, "relerr %.2e" % relerr(a, r), "bad elements:", bad[:6].tolist(), len(bad))

# which op, which family?
from oracle import ref_cpu as R
from cloud_transformers_amd import ops, _lib
lib = _lib.load()
b, h, p = 1, 5, 14498
k1 = keys[b:b + 1, h * dim:(h + 1) * dim].cpu()
f1 = feat[b:b + 1, h * C:(h + 1) * C].cpu()
c1 = cot[b:b + 1, h * C:(h + 1) * C].cpu()
print("key of the point:", k1[0, :, p].tolist(), "scaled:", ((k1[0, :, p] + 1) * (Wn - 1) / 2).tolist())
kk = k1.clone().requires_grad_(True)
lc, idx = R.positions(kk, W, 1, dim)
z = R.splat(lc, idx, f1, None, W, 1, dim, "max")
zc = z.detach().clone().requires_grad_(True)
(gk_splat,) = torch.autograd.grad(z, kk, step.g_z[b:b + 1, h * C:(h + 1) * C].cpu(), retain_graph=True)
kk2 = k1.clone().requires_grad_(True)
lc2, idx2 = R.positions(kk2, W, 1, dim)
o = R.slice_(lc2, idx2, z.detach(), None, W, 1, dim)
(gk_slice,) = torch.autograd.grad(o, kk2, c1)
for fl, name in ((0, "default"), (_lib.DEBUG_NO_HOT, "no_hot"), (_lib.DEBUG_FORCE_HOT, "force_hot")):
    lib.ct_debug_set_flags(fl)
    kd = keys.clone().requires_grad_(True)
    zd = ops.splat_keys(kd, feat, None, W, H, dim, "max")
    (g1,) = torch.autograd.grad(zd, kd, step.g_z)
    t1 = lib.ct_debug_last_launch().decode()
    kd2 = keys.clone().requires_grad_(True)
    od = ops.slice_keys(kd2, step.z, None, W, H, dim)
    (g2,) = torch.autograd.grad(od, kd2, cot)
    t2 = lib.ct_debug_last_launch().decode()
    lib.ct_debug_set_flags(0)
    e1 = (g1[b, h * dim:(h + 1) * dim].cpu() - gk_splat[0]).abs()
    e2 = (g2[b, h * dim:(h + 1) * dim].cpu() - gk_slice[0]).abs()
    print(name, "splat bwd g_keys (%s): max err %.3e at %s | slice bwd g_keys (%s): max err %.3e at %s" % (
        t1, float(e1.max()), divmod(int(e1.argmax()), N), t2, float(e2.max()), divmod(int(e2.argmax()), N)))
print("oracle splat part at the point:", gk_splat[0, :, p].tolist(), " slice part:", gk_slice[0, :, p].tolist())

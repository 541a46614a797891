"""Reproduce one raster fuzz case and explain a mismatch in g_feat: exact ties (two bit-equal winning products in a cell,
or a zero feature against the zero floor) are the only sanctioned difference to the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tests.test_raster_gpu as T
from oracle import ref_cpu as R
from cloud_transformers_amd import ops
cfg = eval(sys.argv[1]) if len(sys.argv) > 1 else (3, 2, 8, 4096, 3, (32, 8, 32), True, 'max')      # a cfg tuple as tools/soak_fuzz.py prints it
print('cfg', cfg)
B, H, C, N, dim, W, use_pad, reduce = cfg
g = torch.Generator().manual_seed((B * 1000003 + H * 10007 + C * 1009 + N * 31 + dim * 7 + sum(W)) % (2 ** 31))
Wl = list(W)
keys0 = torch.tanh(torch.randn(B, H * dim, N, generator=g) * 1.2)
feat0 = torch.randn(B, H * C, N, generator=g)
cot_o = torch.randn(B, H * C, N, generator=g)
pad = (torch.rand(B, N, generator=g) > 0.2).float() if use_pad else None
k = keys0.clone().requires_grad_(True); f = feat0.clone().requires_grad_(True)
lc, idx = R.positions(k, Wl, H, dim)
z_ref = R.splat(lc, idx, f, pad, Wl, H, dim, reduce)
o_ref = R.slice_(lc, idx, z_ref, pad, Wl, H, dim)
(o_ref * cot_o).sum().backward()
kc = keys0.cuda().requires_grad_(True); fc = feat0.cuda().requires_grad_(True)
padc = pad.cuda() if pad is not None else None
z = ops.splat_keys(kc, fc, padc, Wl, H, dim, reduce)
o = ops.slice_keys(kc, z, padc, Wl, H, dim)
(o * cot_o.cuda()).sum().backward()
print("z equal:", torch.equal(z.detach().cpu(), z_ref.detach()))
for name, a, r in (("g_feat", fc.grad.cpu(), f.grad), ("g_keys", kc.grad.cpu(), k.grad), ("out", o.detach().cpu(), o_ref.detach())):
    d = (a - r).abs()
    tol = 1e-4 * float(r.abs().max())
    bad = (d > tol).nonzero()
    print(name, "max err", float(d.max()), "tol", tol, "bad", bad.shape[0])
    for ix in bad[:4]:
        b, ch, n = [int(v) for v in ix]
        h, c = divmod(ch, C) if name != "g_keys" else (ch // dim, None)
        print("   at", (b, ch, n), "got", float(a[b, ch, n]), "ref", float(r[b, ch, n]), "feat", float(feat0[b, ch, n]) if name == "g_feat" else "")
        if name == "g_feat":
            # this point's products per corner vs the cell maxima, and how many points tie there
            cells = idx[b, h, :, n]
            prods = (feat0[b, ch] * (pad[b] if pad is not None else 1.0))[None, :] * lc[b, h].detach()      # [V, N]
            zz = z_ref.detach().reshape(B, H, C, -1)[b, h, c]
            for v in range(1 << dim):
                cell = int(cells[v])
                same = (idx[b, h] == cell) & (prods == zz[cell]) & (zz[cell] > 0)
                print("      corner", v, "cell", cell, "product", float(prods[v, n]), "cell max", float(zz[cell]), "bit-equal winners in cell:", int(same.sum()))

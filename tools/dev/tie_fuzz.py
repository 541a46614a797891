"""Random numbers of constructed exact ties per plane through the hot Splat(max) backward (repair / group redo / plane redo paths)
against the generic kernels (claims everywhere): everything outside the tied pairs must agree, and every pair's summed gradients.
python tools/dev/tie_fuzz.py [cases [first case]]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from cloud_transformers_amd import _lib, ops


def node_key(W):
    hw = np.float32((W - 1) / 2.0)
    for j in range(2, W - 2):
        k = np.float32(2.0 * j / (W - 1) - 1.0)
        for _ in range(8):
            s = np.float32(np.float32(k + np.float32(1.0)) * hw)
            if float(s) == float(j):
                return float(k)
            k = np.nextafter(k, np.float32(1.0 if s < j else -1.0), dtype=np.float32)
    raise AssertionError(W)


SHAPES = [  # dim, W, C, N, B, H
    (2, 32, 16, 4096, 4, 64), (2, 32, 8, 2048, 4, 64), (2, 16, 16, 4096, 8, 16), (2, 64, 16, 4096, 8, 16),
    (3, 8, 32, 4096, 8, 16), (3, 8, 16, 2048, 4, 64), (2, 32, 8, 16384, 2, 128), (2, 16, 16, 4096, 2, 16),
]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    lib = _lib.load()
    rnd = random.Random(5)
    worst = 0.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for case in range(cases):
        dim, W, C, N, B, H = SHAPES[case % len(SHAPES)]
        g = torch.Generator().manual_seed(100 + case)
        keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
        feat = torch.randn(B, H * C, N, generator=g)
        gz = torch.randn(B, H * C, *([W] * dim), generator=g)
        kn = node_key(W)
        nt = rnd.choice([1, 1, 2, 2, 3, 5])
        pairs = []
        used = set()
        skip = case < first
        for t in range(nt):
            same_plane = t > 0 and rnd.random() < 0.6
            b0, h0 = (pairs[0][0], pairs[0][1]) if same_plane else (rnd.randrange(B), rnd.randrange(H))
            c0 = rnd.randrange(C)
            while True:
                p0, p1 = sorted(rnd.sample(range(N), 2))
                if not ({(b0, h0, p0), (b0, h0, p1)} & used):
                    break
            used |= {(b0, h0, p0), (b0, h0, p1)}
            # (the pairs of a plane sit on the SAME node: several ties in one cell, in different channels or the same one)
            for p in (p0, p1):
                keys[b0, h0 * dim:(h0 + 1) * dim, p] = kn
                feat[b0, h0 * C:(h0 + 1) * C, p] = -1.0
            val = 50.0 + 10.0 * t
            feat[b0, h0 * C + c0, p0] = val
            feat[b0, h0 * C + c0, p1] = val
            pairs.append((b0, h0, c0, p0, p1))
        if skip:
            continue
        outs = {}
        for flag in ("FORCE_HOT", "NO_HOT"):
            kd, fd = keys.cuda().requires_grad_(True), feat.cuda().requires_grad_(True)
            lib.ct_debug_set_flags(getattr(_lib, "DEBUG_" + flag))
            try:
                ops.splat_keys(kd, fd, None, [W] * dim, H, dim, "max").backward(gz.cuda())
                tag = lib.ct_debug_last_launch().decode()
            finally:
                lib.ct_debug_set_flags(0)
            outs[flag] = (kd.grad.cpu(), fd.grad.cpu(), tag)
        (gk, gf, tag), (gk0, gf0, tag0) = outs["FORCE_HOT"], outs["NO_HOT"]
        sk, sf = float(gk0.abs().max()), float(gf0.abs().max())
        # fold every tied point's rows of its plane into one row per plane: the sum over all points of a plane is invariant
        keep = torch.ones(B, H, N, dtype=torch.bool)
        for b0, h0, c0, p0, p1 in pairs:
            keep[b0, h0, p0] = keep[b0, h0, p1] = False
        kf = keep[:, :, None, :].expand(B, H, C, N).reshape(B, H * C, N)
        kk = keep[:, :, None, :].expand(B, H, dim, N).reshape(B, H * dim, N)
        pf = ~keep[:, :, None, :].expand(B, H, C, N).reshape(B, H * C, N)          # the constructed pairs' points only
        pk = ~keep[:, :, None, :].expand(B, H, dim, N).reshape(B, H * dim, N)
        # a chance tie elsewhere (two points whose products agree bit for bit: ~1 per 25 M (cell, channel) pairs) may be awarded to
        # different points by the two families: a g_feat row of a plane WITHOUT constructed ties in which exactly two elements differ
        # is that, and the two points are taken out of the comparison
        natural = 0
        df = ((gf - gf0).abs() > 1e-5 * sf)
        rows = (df & kf).sum(-1)
        for b_, r_ in (rows > 0).nonzero().tolist():
            h_ = r_ // C
            if int(rows[b_, r_]) == 2 and not any(pb == b_ and ph == h_ for pb, ph, _, _, _ in pairs):
                for n_ in df[b_, r_].nonzero().flatten().tolist():
                    keep[b_, h_, n_] = False
                natural += 1
        if natural:
            kf = keep[:, :, None, :].expand(B, H, C, N).reshape(B, H * C, N)
            kk = keep[:, :, None, :].expand(B, H, dim, N).reshape(B, H * dim, N)
        e1 = float(((gf - gf0) * kf).abs().max()) / sf
        e2 = float(((gk - gk0) * kk).abs().max()) / sk
        e3 = float(((gf - gf0) * pf).sum(-1).abs().max()) / sf
        e4 = float(((gk - gk0) * pk).sum(-1).abs().max()) / sk
        worst = max(worst, e1, e2, e3, e4)
        ok = e1 <= 1e-5 and e2 <= 1e-5 and e3 <= 1e-4 and e4 <= 1e-4 and not torch.isnan(gf).any() and not torch.isnan(gk).any()
        if os.environ.get("TIE_FUZZ_VERBOSE"):
            print("   pairs (b, h, c, p0, p1):", pairs)
            bad = ((gf - gf0) * kf).abs().amax(-1)
            bi = (bad > 1e-5 * sf).nonzero()
            print("   g_feat rows off outside the pairs (b, row):", bi[:8].tolist(), "H*C row -> h =", [int(r[1]) // C for r in bi[:8]])
        print("case %2d dim%d W%d C%d N%d B%d H%d ties %d %-34s %s  %.1e %.1e %.1e %.1e" % (case, dim, W, C, N, B, H, nt, tag, "ok" if ok else "FAIL", e1, e2, e3, e4) + ("  (%d chance tie(s) awarded differently)" % natural if natural else ""))
        if not ok:
            sys.exit(1)
    print("tie fuzz: %d cases ok, worst relative difference %.2e" % (cases, worst))


if __name__ == "__main__":
    main()

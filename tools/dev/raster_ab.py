"""GPU dev harness for the hot-shape raster kernels: parity against the oracle on small forced-hot
shapes, hot vs generic on the headline shape, and per-pass timings of both families."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from cloud_transformers_amd import _lib, ops
from cloud_transformers_amd.step import SplatSliceStep
from oracle import ref_cpu as R

lib = _lib.load()


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def chain(keys, feat, cot, W, H, dim, reduce, pad=None):
    k = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    z = ops.splat_keys(k, f, pad, W, H, dim, reduce)
    z.retain_grad()
    o = ops.slice_keys(k, z, pad, W, H, dim)
    o.backward(cot)
    return z.detach(), o.detach(), z.grad, f.grad, k.grad


def oracle_chain(keys, feat, cot, W, H, dim, reduce, pad=None):
    k = keys.clone().requires_grad_(True)
    f = feat.clone().requires_grad_(True)
    lc, idx = R.positions(k, W, H, dim)
    z = R.splat(lc, idx, f, pad, W, H, dim, reduce)
    z.retain_grad()
    o = R.slice_(lc, idx, z, pad, W, H, dim)
    o.backward(cot)
    return z.detach(), o.detach(), z.grad, f.grad, k.grad


def small_cases():
    ok = True
    cases = [
        # B, H, C, N, W, pad, dup
        (2, 3, 8, 1024, (32, 32), False, False),
        (1, 2, 16, 4096, (32, 32), False, False),     # QPT = 2
        (1, 2, 4, 8192, (32, 32), False, False),      # QPT = 0 in splat bwd; slice bwd not fused
        (2, 2, 12, 516, (16, 24), True, False),
        (1, 2, 8, 256, (8, 8), False, True),          # exact ties -> claims pass
        (1, 1, 20, 2048, (16, 16), True, False),
        (1, 2, 8, 2052, (32, 32), False, False),      # QPT = 2 with a ragged tail
    ]
    for reduce in ("max", "sum"):
        for (B, H, C, N, W, use_pad, dup) in cases:
            g = torch.Generator().manual_seed(B * 131 + C * 7 + N)
            dim = 2
            keys = torch.tanh(torch.randn(B, H * dim, N, generator=g))
            feat = torch.randn(B, H * C, N, generator=g)
            if dup:
                keys = keys[..., : N // 2].repeat(1, 1, 2)
                feat = feat[..., : N // 2].repeat(1, 1, 2)
            cot = torch.randn(B, H * C, N, generator=g)
            pad = (torch.rand(B, N, generator=g) > 0.2).float() if use_pad else None
            ref = oracle_chain(keys, feat, cot, list(W), H, dim, reduce, pad)
            for flags, name in ((_lib.DEBUG_FORCE_HOT, "hot"), (_lib.DEBUG_NO_HOT, "gen")):
                lib.ct_debug_set_flags(flags)
                got = chain(keys.cuda(), feat.cuda(), cot.cuda(), list(W), H, dim, reduce, None if pad is None else pad.cuda())
                lib.ct_debug_set_flags(0)
                if dup and reduce == "max":
                    # ties: which copy wins is unspecified; the two copies' gradients sum to the tie-free gradient
                    h = N // 2
                    errs = [relerr(got[0], ref[0]), relerr(got[1], ref[1]), relerr(got[2], ref[2])]
                    gf = got[3].cpu()
                    gk = got[4].cpu()
                    # torch's amax backward splits evenly between ties: sums agree
                    errs.append(relerr(gf[..., :h] + gf[..., h:], ref[3][..., :h] + ref[3][..., h:]))
                    errs.append(relerr(gk[..., :h] + gk[..., h:], ref[4][..., :h] + ref[4][..., h:]))
                else:
                    errs = [relerr(x, y) for x, y in zip(got, ref)]
                bad = max(errs) > 1e-4 or (reduce == "max" and not torch.equal(got[0].cpu(), ref[0]))
                ok = ok and not bad
                print("%s %-3s %-3s B%d H%d C%d N%d W%s pad%d dup%d  z %.1e out %.1e g_z %.1e g_feat %.1e g_keys %.1e"
                      % ("FAIL" if bad else "ok  ", reduce, name, B, H, C, N, W, use_pad, dup, *errs), flush=True)
    return ok


def tags_of(step):
    return step.launch_tags()


def headline(C=16, reduce="max"):
    torch.manual_seed(1234)
    B, N, H, W, dim = 8, 4096, 64, 32, 2
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    res = {}
    for flags, name in ((0, "hot"), (_lib.DEBUG_NO_HOT, "gen")):
        lib.ct_debug_set_flags(flags)
        step = SplatSliceStep(keys, feat, cot, W, H, dim, reduce)
        step.run()
        torch.cuda.synchronize()
        print(name, "tags:", tags_of(step))
        step.run()
        torch.cuda.synchronize()
        res[name] = [t.clone() for t in (step.z, step.out, step.g_z, step.g_feat, step.g_keys())]
        # per-pass timings
        times = {}
        for pname in step.PASSES + ("run",):
            fn = getattr(step, pname)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[pname] = e0.elapsed_time(e1) / 30 * 1e3
        alg = step.algorithmic_bytes()
        print("%s C%d %s: " % (name, C, reduce) + "  ".join("%s %.1f us" % kv for kv in times.items())
              + "  | step %.1f%% of 8 TB/s" % (alg["total"] / (times["run"] * 1e-6) / 8e12 * 100), flush=True)
        lib.ct_debug_set_flags(0)
    names = ("z", "out", "g_z", "g_feat", "g_keys")
    print("hot vs generic:", "  ".join("%s %.1e" % (n, relerr(a, b)) for n, a, b in zip(names, res["hot"], res["gen"])))
    # sampled planes against the oracle
    worst = [0.0] * 5
    for (b, h) in ((0, 0), (3, 17), (7, 63), (5, 31)):
        ks = keys[b:b + 1, h * 2:(h + 1) * 2].cpu()
        fs = feat[b:b + 1, h * C:(h + 1) * C].cpu()
        cs = cot[b:b + 1, h * C:(h + 1) * C].cpu()
        ref = oracle_chain(ks, fs, cs, [W, W], 1, 2, reduce)
        got = (res["hot"][0][b:b + 1, h * C:(h + 1) * C], res["hot"][1][b:b + 1, h * C:(h + 1) * C],
               res["hot"][2][b:b + 1, h * C:(h + 1) * C], res["hot"][3][b:b + 1, h * C:(h + 1) * C],
               res["hot"][4][b:b + 1, h * 2:(h + 1) * 2])
        for i in range(5):
            worst[i] = max(worst[i], relerr(got[i], ref[i]))
    print("hot vs oracle planes:", "  ".join("%s %.1e" % kv for kv in zip(names, worst)), flush=True)


if __name__ == "__main__":
    t0 = time.time()
    ok = small_cases()
    print("small cases:", "ALL OK" if ok else "FAILURES", "(%.0f s)" % (time.time() - t0), flush=True)
    headline(16, "max")
    headline(16, "sum")
    headline(4, "max")

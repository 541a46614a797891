"""Sorted-plane raster backward (csrc/ct_raster_sorted.h) against the scatter form and the oracle; A/B timing on one box.

  python tools/dev/sorted_check.py [--time] [--small]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import _lib  # noqa: E402
from cloud_transformers_amd.step import SplatSliceStep  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def per_channel_err(a, b, H, C):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    B = a.shape[0]
    a, b = a.reshape(B, H * C, -1), b.reshape(B, H * C, -1)
    e = (a - b).abs().amax(dim=2) / b.abs().amax(dim=2).clamp_min(1e-30)
    return float(e.max())


def time_pass(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def headline(args):
    lib = _lib.load()
    B, N, H, W, dim, C = args.B, 4096, args.H, args.W, 2, args.C
    for seed in args.seeds:
        torch.manual_seed(seed)
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        step = SplatSliceStep(keys, feat, cot, W, H, dim, "max", plane_sort=True)
        step.splat_fwd(); step.slice_fwd()
        res = {}
        srec = step.sorted
        step.plane_sort(); torch.cuda.synchronize()
        if args.time and srec is not None:
            print("seed %d plane_sort %.1f us" % (seed, time_pass(step.plane_sort)), flush=True)
        for name, fl in (("scatter", _lib.DEBUG_NO_SORTED), ("sorted", _lib.DEBUG_FORCE_SORTED), ("presorted", _lib.DEBUG_FORCE_SORTED)):
            lib.ct_debug_set_flags(fl)
            step.sorted = srec if name == "presorted" else None
            step.g_z.zero_(); step.g_keys_buf.zero_()
            step.slice_bwd()
            torch.cuda.synchronize()
            tag = lib.ct_debug_last_launch().decode()
            res[name] = (step.g_z.clone(), step.g_keys_buf.clone(), tag)
            step.slice_bwd(); torch.cuda.synchronize()
            same = torch.equal(step.g_z, res[name][0]) and torch.equal(step.g_keys_buf, res[name][1])
            t = time_pass(step.slice_bwd) if args.time else float("nan")
            if name == "sorted" and hasattr(lib, "ct_debug_sorted_stamps"):
                import ctypes
                buf = (ctypes.c_ulonglong * 64)()
                lib.ct_debug_sorted_stamps.argtypes = [ctypes.c_void_p]
                step.slice_bwd(); torch.cuda.synchronize()
                lib.ct_debug_sorted_stamps(buf)
                st = [buf[i] for i in range(12)]
                names = ["zero", "hist", "prefix+scan", "ex+K", "ranks+marks", "items", "", "", "loop", "epilogue", ""]
                print("  stamps (cycles from kernel entry of WG0; s_memtime):", " ".join(
                    "%s:%d" % (names[i], st[i + 1] - st[i]) for i in range(10) if st[i + 1] and st[i] and names[i]), flush=True)
                g = [buf[i] for i in range(16, 24)]
                gn = ["wait+max", "stage", "barrier1", "request", "items", "barrier2", "writeout"]
                print("  group 1:", " ".join("%s:%d" % (gn[i], g[i + 1] - g[i]) for i in range(7)), flush=True)
                t0 = min(buf[24 + w] for w in range(16))
                print("  group 1 items per wave (start-t0 .. end-t0):", " ".join(
                    "w%d:%d..%d" % (w, buf[24 + w] - t0, buf[40 + w] - t0) for w in range(16)), flush=True)
            print("seed %d %-8s tag=%s reproducible=%s  %.1f us" % (seed, name, tag, same, t), flush=True)
        lib.ct_debug_set_flags(0)
        step.sorted = srec
        print("  presorted == sorted bit for bit:", torch.equal(res["sorted"][0], res["presorted"][0]) and
              torch.equal(res["sorted"][1], res["presorted"][1]), flush=True)
        print("  sorted vs scatter: g_z per-channel %.2e  g_keys %.2e" % (
            per_channel_err(res["sorted"][0], res["scatter"][0], H, C), relerr(res["sorted"][1], res["scatter"][1])), flush=True)
        # oracle on a few planes
        for (b, h) in ((0, 0), (B // 2, H // 3), (B - 1, H - 1)):
            k = keys[b:b + 1, h * 2:(h + 1) * 2].cpu().clone().requires_grad_(True)
            z = step.z[b:b + 1, h * C:(h + 1) * C].cpu().clone().requires_grad_(True)
            lc, idx = R.positions(k, [W, W], 1, dim)
            o = R.slice_(lc, idx, z, None, [W, W], 1, dim)
            o.backward(cot[b:b + 1, h * C:(h + 1) * C].cpu())
            gz = res["sorted"][0][b:b + 1, h * C:(h + 1) * C]
            gk = res["sorted"][1][b:b + 1, h * 2:(h + 1) * 2]
            print("  plane (%d,%d) vs oracle: g_z per-channel %.2e  g_keys %.2e" % (
                b, h, per_channel_err(gz, z.grad, 1, C), relerr(gk, k.grad)), flush=True)


SMALL = [
    # B, H, C, N, W, pad, dup, nonfinite
    (2, 3, 8, 1024, (32, 32), False, False, False),
    (1, 2, 16, 4096, (32, 32), False, False, False),
    (2, 2, 12, 516, (16, 24), True, False, False),
    (1, 2, 8, 256, (8, 8), False, True, False),
    (1, 1, 20, 2048, (16, 16), True, False, False),
    (1, 2, 8, 2052, (32, 32), False, False, False),
    (1, 2, 8, 4096, (32, 32), False, True, False),      # heavy duplicates: cells with many items
    (1, 2, 8, 1024, (32, 32), False, False, True),      # a channel with inf / NaN
    (1, 1, 8, 4096, (4, 4), False, False, False),       # 9 base cells: items of one cell fill whole waves
]


def small(args):
    from cloud_transformers_amd import ops
    lib = _lib.load()
    worst = 0.0
    for (B, H, C, N, W, pad, dup, nonfinite) in SMALL:
        torch.manual_seed(7)
        dim = len(W)
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda") * (0.3 if dup else 1.0))
        if dup:
            keys[:, :, N // 2:] = keys[:, :, :N // 2]
        if W == (4, 4):
            keys = keys * 0.2
        z = torch.randn(B, H * C, *W, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        if nonfinite:
            cot[0, 1, 5] = float("inf")
            cot[0, 2, 9] = float("nan")
        p = (torch.rand(B, N, device="cuda") > 0.2).float() if pad else None
        out = {}
        for name, fl in (("scatter", _lib.DEBUG_NO_SORTED | _lib.DEBUG_FORCE_HOT), ("sorted", _lib.DEBUG_FORCE_SORTED | _lib.DEBUG_FORCE_HOT)):
            lib.ct_debug_set_flags(fl)
            k = keys.clone().requires_grad_(True)
            zz = z.clone().requires_grad_(True)
            o = ops.slice_keys(k, zz, p, list(W), H, dim)
            o.backward(cot)
            torch.cuda.synchronize()
            out[name] = (zz.grad.clone(), k.grad.clone(), lib.ct_debug_last_launch().decode())
        lib.ct_debug_set_flags(0)
        k = keys.cpu().clone().requires_grad_(True)
        zz = z.cpu().clone().requires_grad_(True)
        lc, idx = R.positions(k, list(W), H, dim)
        o = R.slice_(lc, idx, zz, p.cpu() if pad else None, list(W), H, dim)
        o.backward(cot.cpu())
        if nonfinite:
            fin = torch.isfinite(zz.grad)
            same_nf = bool((torch.isfinite(out["sorted"][0].cpu()) == fin).all())
            e_gz = relerr(torch.where(fin, out["sorted"][0].cpu(), torch.zeros(())), torch.where(fin, zz.grad, torch.zeros(())))
            fk = torch.isfinite(k.grad)
            e_gk = relerr(torch.where(fk, out["sorted"][1].cpu(), torch.zeros(())), torch.where(fk, k.grad, torch.zeros(())))
            print("%s tags %s | %s: non-finite pattern equal %s, finite part g_z %.2e g_keys %.2e" % (
                (B, H, C, N, W), out["scatter"][2], out["sorted"][2], same_nf, e_gz, e_gk), flush=True)
            continue
        e = (per_channel_err(out["sorted"][0], zz.grad, H, C), relerr(out["sorted"][1], k.grad),
             per_channel_err(out["scatter"][0], zz.grad, H, C), relerr(out["scatter"][1], k.grad))
        worst = max(worst, e[0], e[1])
        print("%s pad=%s dup=%s tags %s | %s: sorted vs oracle g_z %.2e g_keys %.2e   (scatter: %.2e %.2e)" % (
            (B, H, C, N, W), pad, dup, out["scatter"][2], out["sorted"][2], *e), flush=True)
    print("worst sorted-vs-oracle error %.2e" % worst)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--C", type=int, default=16)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--W", type=int, default=32)
    ap.add_argument("--seeds", type=int, nargs="*", default=[1234])
    args = ap.parse_args()
    if args.small:
        small(args)
    headline(args)

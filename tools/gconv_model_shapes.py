"""Which grouped 3^d convolutions the three training steps run, and what each costs: one eager fwd+bwd of the classifier
(B8 N2048), the segmenter (B8 N4096) and the inpainter (B2, 2048 -> 16384) with `GroupedConvFn` logging its shapes, then every
distinct shape timed stand-alone through the C ABI (HIP events, us): forward, backward-data, backward-weight; calls per step
and the share of the step's grouped-conv time.    python3 tools/gconv_model_shapes.py [classifier segmenter inpainter]"""
import collections
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd import _lib                                      # noqa: E402
from cloud_transformers_amd.layers import gconv as G                         # noqa: E402
from cloud_transformers_amd.ops import _ptr, _stream                         # noqa: E402

LOG = collections.Counter()
_fwd = G.GroupedConvFn.forward


def _logged(ctx, x, weight, bias, groups):
    LOG[(x.shape[0], groups, x.shape[1] // groups, weight.shape[0] // groups, tuple(x.shape[2:]), bias is not None)] += 1
    return _fwd(ctx, x, weight, bias, groups)


G.GroupedConvFn.forward = staticmethod(_logged)


def run_classifier():
    from tests.test_zoo_gpu import Classifier
    net = Classifier().cuda().train()
    cloud = torch.rand(8, 3, 1, 2048, device="cuda") * 2 - 1
    logits, mask = net(cloud)
    (logits.sum() + mask.sum()).backward()


def run_segmenter():
    from tools.segmenter_step_bench import Segmenter
    net = Segmenter().cuda().train()
    net(torch.rand(8, 6, 4096, device="cuda") * 2 - 1).sum().backward()


def run_inpainter():
    from cloud_transformers_amd.metrics import sphere_noise
    from tests.test_zoo_gpu import Inpainter
    net = Inpainter().cuda().train()
    gen = torch.Generator(device="cuda").manual_seed(1)
    partial = torch.rand(2, 3, 1, 2048, device="cuda", generator=gen) - 0.5
    noise = torch.cat([sphere_noise(2, 16384, "cuda", gen), torch.zeros(2, 1, 16384, device="cuda")], dim=1)
    net(noise, partial).sum().backward()


def t(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def time_shape(B, Gr, Ci, Co, W, bias):
    lib = _lib.load()
    dim = len(W)
    x = torch.randn(B, Gr * Ci, *W, device="cuda")
    w = torch.randn(Gr * Co, Ci, *([3] * dim), device="cuda") * 0.05
    b = torch.randn(Gr * Co, device="cuda")
    y = torch.empty(B, Gr * Co, *W, device="cuda")
    gy = torch.randn_like(y)
    gx, gw, gb = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    Wa = _lib.int_array(W)
    nws = lib.ct_gconv_bwd_weight_workspace_bytes(B, Gr, Ci, Co, dim, Wa)
    ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
    f = t(lambda: _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b) if bias else None, _ptr(y), B, Gr, Ci, Co, dim, Wa, _stream()), "f"))
    d = t(lambda: _lib.check(lib.ct_gconv_bwd_data(_ptr(gy), _ptr(w), _ptr(gx), B, Gr, Ci, Co, dim, Wa, _stream()), "d"))
    g = t(lambda: _lib.check(lib.ct_gconv_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb) if bias else None, _ptr(ws), nws, B, Gr, Ci, Co, dim, Wa, _stream()), "w"))
    return f, d, g


def main():
    torch.manual_seed(0)
    models = [a for a in sys.argv[1:] if not a.startswith("-")] or ["classifier", "segmenter", "inpainter"]
    for name in models:
        LOG.clear()
        {"classifier": run_classifier, "segmenter": run_segmenter, "inpainter": run_inpainter}[name]()
        torch.cuda.synchronize()
        rows = []
        for (B, Gr, Ci, Co, W, bias), n in LOG.items():
            f, d, g = time_shape(B, Gr, Ci, Co, W, bias)
            flop = 2.0 * B * Gr * Ci * Co * 3 ** len(W) * float(torch.tensor(W).prod())
            rows.append((n * (f + d + g), n, B, Gr, Ci, Co, W, f, d, g, flop))
        rows.sort(reverse=True)
        total = sum(r[0] for r in rows)
        print("== %s: %d grouped-conv layers, %d distinct shapes, %.2f ms of fwd + bwd-data + bwd-weight per step" % (name, sum(LOG.values()), len(rows), total / 1e3))
        print("   share  calls  B   G   Cin->Cout  grid         fwd  bwd_data  bwd_weight  (us)   TFLOP/s fwd / data / weight")
        for tot, n, B, Gr, Ci, Co, W, f, d, g, flop in rows:
            print("  %5.1f %%  %3d   %d  %3d  %3d->%-3d   %-10s %6.0f %8.0f %10.0f          %5.1f / %5.1f / %5.1f" % (
                100 * tot / total, n, B, Gr, Ci, Co, "x".join(map(str, W)), f, d, g, flop / f / 1e6, flop / d / 1e6, flop / g / 1e6))
        sys.stdout.flush()


if __name__ == "__main__":
    main()

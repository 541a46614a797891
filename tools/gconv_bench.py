"""Time the MFMA grouped conv against PyTorch/MIOpen on the zoo shapes (us, fwd / fwd+bwd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d


def bench(cls, B, G, C, W, ref):
    m = cls(G * C, G * C, 3, padding=1, groups=G).cuda()
    x = torch.randn(B, G * C, *W, device="cuda", requires_grad=True)
    fn = torch.nn.functional.conv3d if len(W) == 3 else torch.nn.functional.conv2d
    f = (lambda: fn(x, m.weight, m.bias, padding=1, groups=G)) if ref else (lambda: m(x))
    y = f()
    g = torch.randn_like(y)
    for _ in range(20):          # (the small rows are host-bound: the autograd engine's thread and the allocator need more than three rounds to settle)
        y = f(); y.backward(g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        y = f()
    e1.record(); torch.cuda.synchronize(); tf = e0.elapsed_time(e1) / 100
    e0.record()
    for _ in range(100):
        y = f(); y.backward(g)
    e1.record(); torch.cuda.synchronize(); tfb = e0.elapsed_time(e1) / 100
    return round(tf * 1e3), round(tfb * 1e3)


SHAPES = [("2D 32^2 C16 H64", GroupedConv2d, 8, 64, 16, (32, 32)), ("2D 128^2 C4 H16", GroupedConv2d, 8, 16, 4, (128, 128)),
          ("2D 64^2 C16 H16", GroupedConv2d, 8, 16, 16, (64, 64)), ("2D 16^2 C16 H16", GroupedConv2d, 8, 16, 16, (16, 16)),
          ("3D 32^3 C4 H16", GroupedConv3d, 8, 16, 4, (32, 32, 32)), ("3D 16^3 C16 H16", GroupedConv3d, 8, 16, 16, (16, 16, 16)),
          ("3D 8^3 C32 H16", GroupedConv3d, 8, 16, 32, (8, 8, 8))]
for name, cls, B, G, C, W in SHAPES:
    print(name, "mfma fwd/fwd+bwd us", bench(cls, B, G, C, W, False), " miopen", bench(cls, B, G, C, W, True), flush=True)

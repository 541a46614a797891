import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
B, G, W = 8, 16, (32, 32, 32)
x = torch.randn(B, G * 4, *W, device="cuda"); w = torch.randn(G * 4, 4, 3, 3, 3, device="cuda") * 0.1; b = torch.randn(G * 4, device="cuda")
y = torch.empty_like(x); Wa = _lib.int_array(W)
for _ in range(5):
    _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), B, G, 4, 4, 3, Wa, _stream()), "f")
torch.cuda.synchronize()
st = y.view(-1).view(torch.int64)[:2048].cpu()
for blk in range(2):
    o = st[blk * 1024:(blk + 1) * 1024]
    n = int(o[1000])
    t = o[:n * 6].reshape(n, 6).double()
    t0 = t[0, 0]
    print("workgroup", blk, "steps", n, "total cycles", float(t[-1, 5] - t0))
    names = ["wait dma+lgkm", "barrier", "load+store issue", "step (mfma)", "flush", "loop back"]
    d = torch.zeros(n, 6)
    for i in range(n):
        for k in range(5):
            d[i, k] = t[i, k + 1] - t[i, k]
        d[i, 5] = (t[i + 1, 0] - t[i, 5]) if i + 1 < n else 0
    print("   step:", "  ".join("%s" % nm for nm in names))
    for i in range(n):
        print("   %2d: " % i + "  ".join("%8.0f" % v for v in d[i]))
    print("   mean:" + "  ".join("%8.0f" % v for v in d[2:-1].mean(0)))

"""tools/dev/gconv_host.py: host time of the pieces of one eager GroupedConv2d fwd+bwd at the 16^2 C16 H16 shape (the row of
tools/gconv_bench.py that sits at 200 us against MIOpen's 120: host-bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
from cloud_transformers_amd.layers.gconv import GroupedConv2d

B, G, C, W = 8, 16, 16, (16, 16)
m = GroupedConv2d(G * C, G * C, 3, padding=1, groups=G).cuda()
x = torch.randn(B, G * C, *W, device="cuda", requires_grad=True)
y = m(x); g = torch.randn_like(y)
lib = _lib.load()
Wa = _lib.int_array(list(W))
w = m.weight.detach().contiguous(); xd = x.detach()
g_x = torch.empty_like(xd); g_w = torch.empty_like(w); g_b = torch.empty(G * C, device="cuda")
ws_bytes = lib.ct_gconv_bwd_weight_workspace_bytes(B, G, C, C, 2, Wa)
ws = torch.empty(max(ws_bytes, 1), device="cuda", dtype=torch.uint8)


def t(name, fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    tw = (time.perf_counter() - t0) / n
    print("%-46s host %7.1f us   wall %7.1f us" % (name, th * 1e6, tw * 1e6), flush=True)


t("ct_gconv_fwd (C ABI call)", lambda: lib.ct_gconv_fwd(_ptr(xd), _ptr(w), _ptr(m.bias), _ptr(y), B, G, C, C, 2, Wa, _stream()))
t("ct_gconv_bwd_data", lambda: lib.ct_gconv_bwd_data(_ptr(g), _ptr(w), _ptr(g_x), B, G, C, C, 2, Wa, _stream()))
t("ct_gconv_bwd_weight_workspace_bytes", lambda: lib.ct_gconv_bwd_weight_workspace_bytes(B, G, C, C, 2, Wa))
t("ct_gconv_bwd_weight", lambda: lib.ct_gconv_bwd_weight(_ptr(xd), _ptr(g), _ptr(g_w), _ptr(g_b), _ptr(ws), ws_bytes, B, G, C, C, 2, Wa, _stream()))
t("torch.empty ws", lambda: torch.empty(ws_bytes, device="cuda", dtype=torch.uint8))
t("module forward (no grad)", lambda: m(xd))
t("module forward (grad)", lambda: m(x))
t("module forward + backward", lambda: m(x).backward(g))
xr = x.detach().clone().requires_grad_(True)
f = lambda: torch.nn.functional.conv2d(xr, m.weight, m.bias, padding=1, groups=G)
t("MIOpen forward (grad)", f)
t("MIOpen forward + backward", lambda: f().backward(g))
print("ws_bytes", ws_bytes)

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, io, pstats
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    for _ in range(50):
        m(x).backward(g)
    pr.enable()
    for _ in range(500):
        m(x).backward(g)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[2:44]))

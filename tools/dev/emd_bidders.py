"""EMD at B2 n=16384: number of unassigned bidders before every one of the 50 iterations (read from the workspace after runs of
1 .. 50 iterations) and the time of the 50-iteration call; `uniform` clouds or a `blob` against a sphere shell."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
B, n = 2, 16384
mode = sys.argv[1] if len(sys.argv) > 1 else "uniform"
torch.manual_seed(0)
if mode == "uniform":
    a = torch.rand(B, n, 3, device="cuda"); b = torch.rand(B, n, 3, device="cuda")
else:   # a blob against a sphere shell, as early in training
    a = torch.randn(B, n, 3, device="cuda") * 0.1
    b = torch.nn.functional.normalize(torch.randn(B, n, 3, device="cuda"), dim=2) * 0.4
nws = lib.ct_emd_workspace_bytes(B, n)
ws = torch.zeros(nws, device="cuda", dtype=torch.uint8)
dist = torch.empty(B, n, device="cuda"); ass = torch.empty(B, n, device="cuda", dtype=torch.int32)
seg = (B * n * 4 + 255) // 256 * 256
us = []
for k in range(1, 51):
    _lib.check(lib.ct_emd_fwd(_ptr(a), _ptr(b), _ptr(dist), _ptr(ass), _ptr(ws), nws, B, n, ctypes.c_float(0.005), k, _stream()), "emd")
    torch.cuda.synchronize()
    us.append(ws[7 * seg:7 * seg + 8].view(torch.int32).tolist())
print(mode, "U per iteration (batch 0, batch 1):")
print(" ".join("%d/%d" % tuple(u) for u in us))
# timing of the 50-iteration call
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    _lib.check(lib.ct_emd_fwd(_ptr(a), _ptr(b), _ptr(dist), _ptr(ass), _ptr(ws), nws, B, n, ctypes.c_float(0.005), 50, _stream()), "emd")
e1.record(); torch.cuda.synchronize()
print("50 iterations: %.2f ms" % (e0.elapsed_time(e1) / 5))

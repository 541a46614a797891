"""Unassigned bidders per EMD iteration for the completion inpainter's OWN (random-initialised) output against the sphere-shell
ground truth of tools/inpainter_step_bench.py, and the per-call time: what the loss costs inside that training step."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
from cloud_transformers_amd.metrics import sphere_noise
from tests.test_zoo_gpu import Inpainter
lib = _lib.load()
B, n_part, n = 2, 2048, 16384
torch.manual_seed(0)
net = Inpainter().cuda().train()
gen = torch.Generator(device="cuda").manual_seed(1)
partial = torch.rand(B, 3, 1, n_part, device="cuda", generator=gen) - 0.5
gt = torch.nn.functional.normalize(torch.randn(B, n, 3, device="cuda", generator=gen), dim=2) * 0.4
noise = torch.cat([sphere_noise(B, n, "cuda", gen), torch.zeros(B, 1, n, device="cuda")], dim=1)
with torch.no_grad():
    rec = net(noise, partial).squeeze(2).transpose(1, 2).contiguous()
print("rec: mean |x| %.3f  std %.3f   gt std %.3f" % (float(rec.abs().mean()), float(rec.std()), float(gt.std())))
nws = lib.ct_emd_workspace_bytes(B, n)
ws = torch.zeros(nws, device="cuda", dtype=torch.uint8)
dist = torch.empty(B, n, device="cuda"); ass = torch.empty(B, n, device="cuda", dtype=torch.int32)
seg = (B * n * 4 + 255) // 256 * 256
us = []
for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 40, 50):
    _lib.check(lib.ct_emd_fwd(_ptr(rec), _ptr(gt), _ptr(dist), _ptr(ass), _ptr(ws), nws, B, n, ctypes.c_float(0.005), k, _stream()), "emd")
    torch.cuda.synchronize()
    us.append((k, ws[7 * seg:7 * seg + 8].view(torch.int32).tolist()))
print("U before iteration k:", " ".join("%d:%d/%d" % (k, u[0], u[1]) for k, u in us))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    _lib.check(lib.ct_emd_fwd(_ptr(rec), _ptr(gt), _ptr(dist), _ptr(ass), _ptr(ws), nws, B, n, ctypes.c_float(0.005), 50, _stream()), "emd")
e1.record(); torch.cuda.synchronize()
print("50 iterations: %.2f ms" % (e0.elapsed_time(e1) / 5))
# determinism and oracle check on this collapsed cloud
outs = []
for _ in range(3):
    _lib.check(lib.ct_emd_fwd(_ptr(rec), _ptr(gt), _ptr(dist), _ptr(ass), _ptr(ws), nws, B, n, ctypes.c_float(0.005), 50, _stream()), "emd")
    torch.cuda.synchronize()
    outs.append((dist.clone(), ass.clone()))
print("repeatable:", all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:]), " mean dist %.6f" % float(outs[0][0].mean()))
try:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
    import emd_ref
    import numpy as np
    st, d_ref, a_ref = emd_ref.forward(rec[:1].cpu().numpy(), gt[:1].cpu().numpy(), 0.005, 8)
    _lib.check(lib.ct_emd_fwd(_ptr(rec), _ptr(gt), _ptr(dist), _ptr(ass), _ptr(ws), nws, B, n, ctypes.c_float(0.005), 8, _stream()), "emd")
    torch.cuda.synchronize()
    print("oracle (batch 0, 8 iterations): assignment equal", bool(np.array_equal(ass[0].cpu().numpy(), a_ref[0])), " dist equal", bool(np.array_equal(dist[0].cpu().numpy(), d_ref[0])))
except Exception as e:
    print("oracle check skipped:", str(e)[:200])

"""Per-pass device times of the op-level step at the headline shape (or --feat/--reduce), for the library in CLOUDCT_LIB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from cloud_transformers_amd import _lib
from cloud_transformers_amd.step import SplatSliceStep

C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reduce = sys.argv[2] if len(sys.argv) > 2 else "max"
torch.manual_seed(1234)
B, N, H, W, dim = 8, 4096, 64, 32, 2
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
step = SplatSliceStep(keys, feat, cot, W, H, dim, reduce)
step.run()
torch.cuda.synchronize()
times = {}
for rep in range(2):
    for pname in step.PASSES + ("run",):
        fn = getattr(step, pname)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[pname] = min(times.get(pname, 1e9), e0.elapsed_time(e1) / 50 * 1e3)
alg = step.algorithmic_bytes()
print(os.path.basename(_lib.LIB_PATH), "C%d %s:" % (C, reduce), "  ".join("%s %.1f" % kv for kv in times.items()),
      "| step %.1f%%" % (alg["total"] / (times["run"] * 1e-6) / 8e12 * 100), flush=True)

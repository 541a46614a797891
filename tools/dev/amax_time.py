"""us per ct_amax_f32 call on the model's tensor sizes (HIP events over 50 calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import ops

for shape in [(848, 512), (8, 512, 4096), (8, 848, 4096), (8, 512, 2048), (2, 512, 16384)]:
    x = torch.randn(*shape, device="cuda")
    for _ in range(5):
        ops.amax(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.amax(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    assert float(ops.amax(x).max()) == float(x.abs().max())
    print(shape, "%.1f us  %.2f TB/s" % (us, x.numel() * 4 / us * 1e-6))

"""Splat(max) backward time by point segments (ct_debug_set_nseg) on the zoo head shapes: 1 = chunk groups (+ fold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep

SHAPES = [(16, 64, 2), (16, 16, 3), (16, 16, 2), (32, 8, 3)]
for B, N in [(8, 4096), (8, 2048), (2, 16384)]:
    for C, W, dim in SHAPES:
        torch.manual_seed(0)
        H = 16
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
        st.run()
        row = []
        for ns in (1, 2, 4, 8, 16):
            st.lib.ct_debug_set_nseg(ns)
            for _ in range(20):
                st.splat_bwd()
            tag = st.lib.ct_debug_last_launch().decode()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                st.splat_bwd()
            e1.record()
            torch.cuda.synchronize()
            row.append("%d:%s%.1f" % (ns, "s" if "segments" in tag else "g", e0.elapsed_time(e1) * 10))
        st.lib.ct_debug_set_nseg(0)
        print("C%d W%d d%d B%d N%d |" % (C, W, dim, B, N), " ".join(row), flush=True)

"""launch tags of the four passes on the zoo head shapes, with and without the arrival tickets (tools/zoo_sweep.py's two modes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
SHAPES = [(4, 128, 2), (4, 32, 3), (16, 64, 2), (16, 16, 3), (16, 16, 2), (32, 8, 3)]
for B, N in [(8, 4096), (8, 2048), (2, 16384)]:
    for C, W, dim in SHAPES:
        torch.manual_seed(0)
        H = 16
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        for tk in (True, False):
            st = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=tk)
            st.run(); torch.cuda.synchronize()
            t = st.launch_tags()
            print(C, W, dim, "|", B, N, "| tickets" if tk else "| plain  ", "|", t.get("slice_bwd"), "|", t.get("splat_bwd"))

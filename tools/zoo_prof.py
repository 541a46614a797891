"""The six zoo head shapes (segmenter.py:28-45, inpainter.py:135-155) as plain launches, for rocprofv3 passes:
every shape's launches are preceded by ONE launch of the occupancy kernel (the marker tools/zoo_prof_report.py cuts the
dispatch sequence at), so kernel-trace durations, SQ counters and FETCH/WRITE_SIZE of template instances shared between
shapes can be attributed per shape.

    python3 tools/zoo_prof.py [B8N4096] [B2N16384] [B8N2048]      (default: the first two)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd import ops
from cloud_transformers_amd.step import SplatSliceStep

SHAPES = [(4, 128, 2), (4, 32, 3), (16, 64, 2), (16, 16, 3), (16, 16, 2), (32, 8, 3)]
CONFIGS = {"B8N4096": (8, 4096), "B2N16384": (2, 16384), "B8N2048": (8, 2048)}
ITERS = 12


def main():
    names = [a for a in sys.argv[1:] if a in CONFIGS] or ["B8N4096", "B2N16384"]
    marker = torch.ones(64, device="cuda")
    for name in names:
        B, N = CONFIGS[name]
        for C, W, dim in SHAPES:
            torch.manual_seed(0)
            H = 16
            keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
            feat = torch.randn(B, H * C, N, device="cuda")
            cot = torch.randn(B, H * C, N, device="cuda")
            st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
            torch.cuda.synchronize()
            ops.grid_occupancy_count(marker)          # the marker launch
            for _ in range(ITERS):
                st.run()
            torch.cuda.synchronize()
    print("order:", [(n, s) for n in names for s in SHAPES])


if __name__ == "__main__":
    main()

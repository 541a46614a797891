"""tools/dev/gconv_small.py [W=16] [C=16] [dim=2]: one small grouped conv (B8, 16 groups) fwd+bwd x 20 — for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cls = GroupedConv2d if dim == 2 else GroupedConv3d
m = cls(16 * C, 16 * C, 3, padding=1, groups=16).cuda()
x = torch.randn(8, 16 * C, *([W] * dim), device="cuda", requires_grad=True)
y = m(x)
g = torch.randn_like(y)
for _ in range(20):
    y = m(x); y.backward(g)
torch.cuda.synchronize()

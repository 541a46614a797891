import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d
cfg = sys.argv[1] if len(sys.argv) > 1 else "2d"
if cfg == "2d":
    m = GroupedConv2d(64*16, 64*16, 3, padding=1, groups=64).cuda(); x = torch.randn(8, 64*16, 32, 32, device="cuda", requires_grad=True)
else:
    m = GroupedConv3d(16*16, 16*16, 3, padding=1, groups=16).cuda(); x = torch.randn(8, 16*16, 16, 16, 16, device="cuda", requires_grad=True)
y = m(x); g = torch.randn_like(y)
for _ in range(10):
    y = m(x); y.backward(g)
torch.cuda.synchronize()

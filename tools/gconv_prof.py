import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.layers.gconv import GroupedConv2d, GroupedConv3d
cfg = sys.argv[1] if len(sys.argv) > 1 else "2d"
WIDE = {"8c64": (3, 64, 8), "4c64": (3, 64, 4), "2c64": (3, 64, 2), "8c64_2d": (2, 64, 8), "8c32": (3, 32, 8), "32c4": (3, 4, 32)}      # (dim, channels per group, extent): the Res stacks
if cfg in WIDE:
    d, c, w = WIDE[cfg]
    cls = GroupedConv3d if d == 3 else GroupedConv2d
    m = cls(16 * c, 16 * c, 3, padding=1, groups=16, bias=False).cuda()
    x = torch.randn(8, 16 * c, *([w] * d), device="cuda", requires_grad=True)
elif cfg == "2d":
    m = GroupedConv2d(64*16, 64*16, 3, padding=1, groups=64).cuda(); x = torch.randn(8, 64*16, 32, 32, device="cuda", requires_grad=True)
else:
    m = GroupedConv3d(16*16, 16*16, 3, padding=1, groups=16).cuda(); x = torch.randn(8, 16*16, 16, 16, 16, device="cuda", requires_grad=True)
y = m(x); g = torch.randn_like(y)
for _ in range(10):
    y = m(x); y.backward(g)
torch.cuda.synchronize()

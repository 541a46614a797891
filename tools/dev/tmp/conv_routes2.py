import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_zoo_gpu as T
from cloud_transformers_amd.layers import gconv as GC
net = T.Classifier().cuda().train()
seen = []
def hook(m, a):
    x = a[0]
    seen.append((type(m).__name__, tuple(x.shape), m.in_channels, m.out_channels, m.groups, tuple(m.kernel_size), GC._eligible(m, x), GC._pointwise(m, x)))
for m in net.modules():
    if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)):
        m.register_forward_pre_hook(hook)
cloud = torch.rand(8, 3, 1, 2048, device="cuda") * 2 - 1
net(cloud)
for s in seen:
    if not s[6]:
        print(s)
print(len(seen), "conv calls;", sum(1 for s in seen if s[6]), "on ct_gconv")

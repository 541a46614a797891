"""Which convolution / linear modules of a zoo model run on the library (MIOpen / hipBLASLt) and which on our kernels:
one forward of the classifier with hooks, prints (module kind, groups, cin/g, cout/g, kernel, spatial) -> count, route."""
import collections
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_zoo_gpu import Classifier          # noqa: E402
from cloud_transformers_amd.layers import gconv    # noqa: E402

net = Classifier().cuda().train()
seen = collections.Counter()


def hook(mod, inp, out):
    x = inp[0]
    if isinstance(mod, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
        own = isinstance(mod, (gconv.GroupedConv2d, gconv.GroupedConv3d)) and gconv._eligible(mod, x)
        key = (type(mod).__name__, "g%d" % mod.groups, mod.in_channels // mod.groups, mod.out_channels // mod.groups,
               tuple(mod.kernel_size), tuple(x.shape), "own" if own else "LIB")
    else:
        key = (type(mod).__name__, tuple(x.shape), tuple(out.shape))
    seen[key] += 1


for m in net.modules():
    if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.Linear)):
        m.register_forward_hook(hook)
cloud = torch.rand(8, 3, 1, 2048, device="cuda") * 2 - 1
net(cloud)
for k, v in sorted(seen.items(), key=lambda kv: str(kv[0])):
    print(v, k)

"""Which Python lines launch the small fill / copy / mul kernels of a block step (torch.profiler with stacks, one eager fwd+bwd of a
MultiHeadUnion block)."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd.layers.multihead_ct import MultiHeadUnion

torch.manual_seed(0)
blk = MultiHeadUnion(512, [16, 16], [64, 16], [2, 3], [16, 16], model_dim_out=512).cuda()
x = torch.randn(8, 512, 4096, device="cuda", requires_grad=True)
xyz = torch.rand(8, 3, 4096, device="cuda") * 2 - 1
for _ in range(2):
    y, _ = blk(x, xyz)
    y.sum().backward()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    y, _ = blk(x, xyz)
    y.sum().backward()
cnt = Counter()
for ev in prof.events():
    if ev.name in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::zeros_like", "aten::copy_", "aten::mul", "aten::add", "aten::contiguous", "aten::clone"):
        st = [s for s in ev.stack if "cloud_transformers_amd" in s or "tools/" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, where), n in cnt.most_common(40):
    print(n, name, where)

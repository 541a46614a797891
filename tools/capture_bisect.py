"""HIP-graph capture check, one subprocess per case (a capture bug in the stack below would take the
process down, so this is a tool, not a pytest test): each libcloudct op and whole MultiHeadUnion blocks
under torch.cuda.make_graphed_callables.  All cases pass on ROCm 7.0 / MI355X."""
import sys, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = ["small_eval", "small_train", "mid_train", "splat_only", "slice_only", "gconv_only", "lattice_only"]
if len(sys.argv) == 1:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=180)
        print(v, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:100])
    sys.exit(0)
import torch
from cloud_transformers_amd.layers.multihead_ct import MultiHeadUnion
from cloud_transformers_amd import ops
v = sys.argv[1]
torch.manual_seed(0)
class OnlyOut(torch.nn.Module):
    def __init__(self, inner): super().__init__(); self.inner = inner
    def forward(self, a, b): return self.inner(a, b)[0]
class Fn(torch.nn.Module):
    def __init__(self, f): super().__init__(); self.f = f; self.p = torch.nn.Parameter(torch.ones(1))
    def forward(self, a, b): return self.f(a * self.p, b)
if v in ("small_eval", "small_train"):
    m = OnlyOut(MultiHeadUnion(32, [4, 4], [16, 8], [2, 3], [4, 2])).cuda()
    m.train(v == "small_train")
    x = torch.randn(2, 32, 256, device="cuda", requires_grad=True); y = torch.rand(2, 3, 256, device="cuda") * 2 - 1
elif v == "mid_train":
    m = OnlyOut(MultiHeadUnion(64, [4, 4], [16, 8], [2, 3], [8, 8])).cuda().train()
    x = torch.randn(4, 64, 1024, device="cuda", requires_grad=True); y = torch.rand(4, 3, 1024, device="cuda") * 2 - 1
elif v == "splat_only":
    m = Fn(lambda f, k: ops.splat_keys(k, f, None, [16, 16], 4, 2)).cuda()
    x = torch.randn(2, 16, 256, device="cuda", requires_grad=True); y = torch.tanh(torch.randn(2, 8, 256, device="cuda"))
elif v == "slice_only":
    m = Fn(lambda g, k: ops.slice_keys(k, g, None, [16, 16], 4, 2)).cuda()
    x = torch.randn(2, 16, 16, 16, device="cuda", requires_grad=True); y = torch.tanh(torch.randn(2, 8, 256, device="cuda"))
elif v == "gconv_only":
    from cloud_transformers_amd.layers.gconv import GroupedConv2d
    c = GroupedConv2d(16, 16, 3, padding=1, groups=4).cuda()
    m = Fn(lambda a, b: c(a)).cuda(); m.c = c
    x = torch.randn(2, 16, 16, 16, device="cuda", requires_grad=True); y = torch.zeros(1, device="cuda")
elif v == "lattice_only":
    from cloud_transformers_amd.layers.utils import so3_exponential_map
    R = so3_exponential_map(torch.randn(4, 3, device="cuda")); sh = torch.zeros(4, 3, device="cuda")
    m = Fn(lambda res, xyz: ops.lattice(xyz, res, R, sh, None, None, 2)[1]).cuda()
    x = torch.randn(2, 12, 256, device="cuda", requires_grad=True); y = torch.rand(2, 3, 256, device="cuda")
gm = torch.cuda.make_graphed_callables(m, (x, y))
out = gm(x, y); out.sum().backward(); torch.cuda.synchronize()
print("ok", float(out.sum()))

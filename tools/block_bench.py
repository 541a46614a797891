"""Block-level benchmark: one MultiHeadUnion (2D 64^2 + 3D 16^3 heads, C=16, H=16 each,
model_dim 512) forward+backward at the S3DIS per-GPU shape (B=8, N=4096), on one MI355X:
  ours   — cloud_transformers_amd modules (HIP Splat/Slice, MFMA grouped conv)
  eager  — the reference FORMULATION in eager PyTorch-ROCm on the same GPU: materialised
           (B,H,C,V,N) pre_splat, int64 indices expanded over C, scatter_reduce(amax) / gather,
           MIOpen grouped conv (what running the reference's Python on this GPU would execute,
           with torch's scatter_reduce standing in for torch_scatter)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from cloud_transformers_amd.layers import multihead_ct as M
from cloud_transformers_amd.layers.cloud_transform import Splat, Slice


def eager_positions(keys, W, H, dim):
    B, _, N = keys.shape
    k = keys.reshape(B * H, dim, N).clamp(-1 + 1e-7, 1 - 1e-7)
    mod = torch.tensor(W, dtype=torch.float32, device=keys.device)[None, :, None]
    s = (k + 1.0) * ((mod - 1) * 0.5)
    s = k + (s - k).detach() if False else s          # (grad balancing ignored for timing purposes)
    f = s.floor()
    w0, w1 = (f + 1) - s, s - f
    ws, cells = [], []
    for v in range(1 << dim):
        off = [(v >> j) & 1 for j in range(dim)]
        wv = None
        for j in range(dim):
            wj = w1[:, j] if off[j] else w0[:, j]
            wv = wj if wv is None else wv * wj
        ws.append(wv)
        c = [f[:, j].long() + off[j] for j in range(dim)]
        cells.append(c[0] * W[1] + c[1] if dim == 2 else c[0] * W[1] * W[2] + c[1] * W[2] + c[2])
    V = 1 << dim
    return torch.stack(ws, 1).reshape(B, H, V, N), torch.stack(cells, 1).reshape(B, H, V, N)


class EagerSplat(Splat):
    def forward_keys(self, keys, features, pad=None):
        W, H, dim = self.tensor_size, self.heads, self.dim
        B, HC, N = features.shape
        C = HC // H
        lc, idx = eager_positions(keys, W, H, dim)
        self._cache = (lc, idx)
        pre = features.reshape(B, H, C, N)[:, :, :, None] * lc[:, :, None]
        G = 1
        for w in W:
            G *= w
        z0 = torch.zeros(B, H, C, G, device=features.device)
        index = idx[:, :, None].reshape(B, H, 1, -1).expand(B, H, C, -1)
        z = z0.scatter_reduce(3, index, pre.reshape(B, H, C, -1), reduce="amax", include_self=True)
        return z.reshape(B, HC, *W)


class EagerSlice(Slice):
    def forward_keys(self, keys, grid, pad=None):
        W, H, dim = self.tensor_size, self.heads, self.dim
        lc, idx = eager_positions(keys, W, H, dim)
        B, _, V, N = lc.shape
        C = grid.shape[1] // H
        index = idx[:, :, None].expand(-1, -1, C, -1, -1).reshape(B, H, C, -1)
        g = torch.gather(grid.reshape(B, H, C, -1), 3, index).reshape(B, H, C, V, N)
        return (g * lc[:, :, None]).sum(3).reshape(B, H * C, N)


def make(eager):
    torch.manual_seed(0)
    m = M.MultiHeadUnion(512, [16, 16], [64, 16], [2, 3], [16, 16]).cuda()
    if eager:
        for att in m.attentions:
            att.splat.__class__ = EagerSplat
            att.slice.__class__ = EagerSlice
            conv = att.conv[0]
            plain = (nn.Conv3d if att.tensor_dim == 3 else nn.Conv2d)(conv.in_channels, conv.out_channels, 3, padding=1,
                                                                     groups=conv.groups).cuda()
            plain.load_state_dict(conv.state_dict())
            att.conv[0] = plain
            att._occupancy = lambda z, batch, a=att: (z.abs() > 1e-9).sum().float() / (batch * a.in_feature_dim * a.heads)
    return m


def run(m, x, pcd, iters=10):
    for _ in range(3):
        out, _ = m(x, pcd); out.square().mean().backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.reset_peak_memory_stats()
    e0.record()
    for _ in range(iters):
        out, _ = m(x, pcd); out.square().mean().backward()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, torch.cuda.max_memory_allocated() / 2**30


B, N = 8, 4096
torch.manual_seed(1)
x = torch.randn(B, 512, N, device="cuda", requires_grad=True)
pcd = torch.rand(B, 3, N, device="cuda") * 2 - 1
ours, mem_o = run(make(False), x, pcd)
eager, mem_e = run(make(True), x, pcd)
print("MultiHeadUnion fwd+bwd B8 N4096: ours %.2f ms (%.0f k points/s, peak %.2f GiB) | eager PyTorch-ROCm formulation %.2f ms (%.0f k points/s, peak %.2f GiB) | speed-up %.1fx"
      % (ours, B * N / ours, mem_o, eager, B * N / eager, mem_e, eager / ours))


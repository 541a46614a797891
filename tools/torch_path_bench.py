"""The reference's own FORMULATION of the hot path — DifferentiablePositions -> Splat -> Slice as the tensor-op sequence of
layers/cloud_transform.py:72-227 (materialised (B,H,C,V,N) pre_splat, expanded int64 gather index; torch's scatter_reduce(amax) where
the reference calls torch_scatter.scatter_max) — executed by eager PyTorch-ROCm ON THE SAME MI355X, beside this library's
kernels: what a user of the reference gets by running its Python on this GPU unchanged.  fwd+bwd per step, HIP-event time.
(The op sequence is restated here on the device; the CPU oracle of the tests is oracle/ref_cpu.py.)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.step import SplatSliceStep


class _Balance(torch.autograd.Function):          # layers/cloud_transform.py:12-31 (GradientBalancing)
    @staticmethod
    def forward(ctx, x, scale):
        return x * scale

    @staticmethod
    def backward(ctx, g):
        return g, None


def torch_path_step(keys, feat, cot, W, H, dim):
    B, _, N = keys.shape
    C = feat.shape[1] // H
    G = math.prod(W)
    keys = keys.detach().clone().requires_grad_(True)
    feat = feat.detach().clone().requires_grad_(True)
    k = keys.reshape(B * H, dim, N).clamp(-1 + 1e-7, 1 - 1e-7)
    mod = torch.tensor(W, dtype=torch.float32, device=keys.device)[None, :, None]
    s = _Balance.apply(k + 1.0, (mod - 1) * 0.5)
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
        cells.append(c[0] * W[1] * W[2] + c[1] * W[2] + c[2] if dim == 3 else c[0] * W[1] + c[1])
    V = 1 << dim
    lc = torch.stack(ws, dim=1).reshape(B, H, V, N)
    idx = torch.stack(cells, dim=1).reshape(B, H, V, N)
    pre = feat.reshape(B, H, C, N)[:, :, :, None] * lc[:, :, None]
    z0 = torch.zeros(B, H, C, G, device=keys.device)
    index = idx[:, :, None].reshape(B, H, 1, -1).expand(B, H, C, -1)
    z = z0.scatter_reduce(3, index, pre.reshape(B, H, C, -1), reduce="amax", include_self=True)
    gi = idx[:, :, None].expand(-1, -1, C, -1, -1).reshape(B, H, C, -1)
    out = (torch.gather(z, 3, gi).reshape(B, H, C, V, N) * lc[:, :, None]).sum(dim=3).reshape(B, H * C, N)
    out.backward(cot)
    return out.detach(), feat.grad, keys.grad


def timeit(f, iters):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


SHAPES = [("headline", 8, 4096, 64, 16, (32, 32)), ("zoo 128^2 C4", 8, 4096, 16, 4, (128, 128)), ("zoo 32^3 C4", 8, 4096, 16, 4, (32, 32, 32)),
          ("zoo 64^2 C16", 8, 4096, 16, 16, (64, 64)), ("zoo 16^3 C16", 8, 4096, 16, 16, (16, 16, 16)), ("zoo 16^2 C16", 8, 4096, 16, 16, (16, 16)),
          ("zoo 8^3 C32", 8, 4096, 16, 32, (8, 8, 8))]
print("shape | eager PyTorch-ROCm op sequence (ms per fwd+bwd) | this library (ms) | ratio | elements of out / g_feat / g_keys further than 1e-4 of "
      "the tensor's max from the op sequence's (exact ties only: torch's amax backward SPLITS a tied cell's cotangent, torch_scatter.scatter_max "
      "and this library award one winner — tests/test_tie_rule_gpu.py)")
for name, B, N, H, C, W in SHAPES:
    dim = len(W)
    torch.manual_seed(0)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    st = SplatSliceStep(keys, feat, cot, list(W), H, dim, "max")
    for _ in range(50):
        st.run()
    t_mine = timeit(st.run, 100)
    o, gf, gk = torch_path_step(keys, feat, cot, list(W), H, dim)
    t_torch = timeit(lambda: torch_path_step(keys, feat, cot, list(W), H, dim), 5)
    st.run(); torch.cuda.synchronize()
    off = ["%d of %d" % (int(((a - b).abs() > 1e-4 * b.abs().max()).sum()), b.numel()) for a, b in ((st.out, o), (st.g_feat, gf), (st.g_keys(), gk))]
    print("%-13s B%d N%d H%d | %8.2f | %7.4f | x%5.0f | %s | %s | %s" % (name, B, N, H, t_torch, t_mine, t_torch / t_mine, *off), flush=True)

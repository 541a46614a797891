"""1x1 Conv1d (the point-wise projections of the MHCT blocks) fwd+bwd: MIOpen's own backward vs a
rocBLAS batched-GEMM weight gradient (no NHWC transposes).  µs per call, B8 N4096."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class PW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.conv1d(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = F.conv1d(gy, w.transpose(0, 1).contiguous()) if ctx.needs_input_grad[0] else None
        gw = torch.bmm(gy, x.transpose(1, 2)).sum(0).unsqueeze(-1) if ctx.needs_input_grad[1] else None
        return gx, gw


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    B, N = 8, 4096
    for O, I in [(304, 512), (512, 512), (512, 256), (128, 512)]:
        x = torch.randn(B, I, N, device="cuda", requires_grad=True)
        w = torch.randn(O, I, 1, device="cuda", requires_grad=True)
        gy = torch.randn(B, O, N, device="cuda")

        def ref():
            x.grad = None; w.grad = None
            F.conv1d(x, w).backward(gy)

        def ours():
            x.grad = None; w.grad = None
            PW.apply(x, w).backward(gy)
        ref(); gr = w.grad.clone(); gxr = x.grad.clone()
        ours()
        err = float((w.grad - gr).abs().max() / gr.abs().max()), float((x.grad - gxr).abs().max() / gxr.abs().max())
        print(f"O{O} I{I}: miopen {timeit(ref):7.1f} us | gemm wgrad {timeit(ours):7.1f} us | rel err gw {err[0]:.1e} gx {err[1]:.1e}", flush=True)


if __name__ == "__main__":
    main()

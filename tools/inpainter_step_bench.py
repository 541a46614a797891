"""A full training step of the completion (inpainting) network at BASELINE config 4's shapes: encoder on a partial cloud
(B2, N2048: stem + 12 MultiHeadUnion blocks + the 2D/3D MultiHeadPool heads with their Res2D/Res3D grouped-conv stacks),
mapping to the style vector, AdaIN decoder on a 16384-point noise cloud (stem + 12 MultiHeadUnionAdaIn blocks + head), and
the training loss of train_inpainter.py:186-192: sqrt(EMD dist) (eps 0.005, 50 auction iterations) + Chamfer; backward,
Adam step.  The network is the tests' restatement of model_zoo/completion/inpainter.py (tests/test_zoo_gpu.py::Inpainter,
whose outputs are pinned on the reference's); synthetic clouds, random-initialised weights.  Prints ms per step with eager
launches and with forward + loss + backward replayed as one HIP graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_transformers_amd.layers.pointwise import convert_pointwise          # noqa: E402
from cloud_transformers_amd.chamfer import loss_chamfer                    # noqa: E402
from cloud_transformers_amd.emd import emdModule                           # noqa: E402
from cloud_transformers_amd.metrics import sphere_noise                    # noqa: E402
from tests.test_zoo_gpu import Inpainter                                   # noqa: E402


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B, n_part, n_out = 2, 2048, 16384
    torch.manual_seed(0)
    net = convert_pointwise(Inpainter().cuda()).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    emd = emdModule()
    gen = torch.Generator(device="cuda").manual_seed(1)
    partial = (torch.rand(B, 3, 1, n_part, device="cuda", generator=gen) - 0.5)
    gt = torch.nn.functional.normalize(torch.randn(B, n_out, 3, device="cuda", generator=gen), dim=2) * 0.4
    gt4 = gt.transpose(1, 2).unsqueeze(2).contiguous()                         # [B, 3, 1, n], as the training script holds it
    noise = torch.cat([sphere_noise(B, n_out, "cuda", gen), torch.zeros(B, 1, n_out, device="cuda")], dim=1)   # [B, 4, n]: xyz + flag
    nparam = sum(p.numel() for p in net.parameters())

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)       # fresh gradients each step (the graph re-creates them in place): no zero fills, no accumulation adds
        rec4 = net(noise, partial)                                             # [B, 3, 1, n]
        rec = rec4.squeeze(2).transpose(1, 2).contiguous()                     # [B, n, 3]
        dist, _ = emd(rec, gt, 0.005, 50)
        loss = torch.sqrt(dist).mean(1).mean() + loss_chamfer(rec4, gt4)
        loss.backward()
        return loss

    def step():
        fwd_bwd()
        opt.step()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    eager = timeit(step, 5)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
    g.replay()

    def graphed_step():
        g.replay()
        opt.step()
    graphed = timeit(graphed_step, 5)
    print(f"completion inpainter ({nparam / 1e6:.1f} M parameters: encoder 12 MultiHeadUnion + pools, decoder 12 MultiHeadUnionAdaIn), "
          f"B{B} partial N{n_part} -> {n_out} points, EMD(50 it) + Chamfer loss, 1x MI355X, fp32: training step eager {eager:.1f} ms | "
          f"fwd+loss+bwd as one HIP graph + optimizer {graphed:.1f} ms ({B * n_out / graphed:.0f} k output points/s) | "
          f"loss {float(static_loss.detach()):.4f}")


if __name__ == "__main__":
    main()

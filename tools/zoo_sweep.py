"""Secondary sweep: the six zoo head shapes at B8 H16, N in {2048, 4096}, plus N=16384 at B2:
per-pass device time and fraction of the 8 TB/s roofline for the fwd+bwd step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes_back_to_back as time_passes

SHAPES = [(4, 128, 2), (4, 32, 3), (16, 64, 2), (16, 16, 3), (16, 16, 2), (32, 8, 3)]
print("C W dim | B N | mode | us per pass ... (each pass back to back with itself) | step us = their sum | algorithmic MB | frac of 8 TB/s | "
      "the four passes as ONE HIP graph, 10 steps per graph (how bench.py times the headline step): us per step | frac")


def graphed_step_us(st, steps_per_graph=10, replays=30):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(steps_per_graph):
            st.run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * steps_per_graph)


# both modes per shape, one after the other on this box (boxes differ by up to 15 %: never compare rows of different runs):
#   tickets = the round-4 path (arrival tickets: partial sums folded inside the producing kernels, point segments for Splat(max)
#   backward, XCD-aware placement); plain = the same entry points without the ticket buffer (the round-3 launches + sum_parts)
MODES = [("tickets", True)] if "--tickets-only" in sys.argv else [("plain", False)] if "--no-tickets" in sys.argv else [("tickets", True), ("plain", False)]
for B, N in [(8, 4096), (8, 2048), (2, 16384)]:
    for C, W, dim in SHAPES:
        torch.manual_seed(0)
        H = 16
        keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
        feat = torch.randn(B, H * C, N, device="cuda")
        cot = torch.randn(B, H * C, N, device="cuda")
        for name, tk in MODES:
            st = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=tk)
            for _ in range(100):            # sustained load first: a short burst after idle runs at ramping clocks (HISTORY.md §5)
                st.run()
            torch.cuda.synchronize()
            p = time_passes(st, iters=100)
            tot = sum(p.values()) * 1e3
            alg = st.algorithmic_bytes()["total"]
            try:
                tg = graphed_step_us(st)
            except Exception as ex:          # noqa: BLE001 — a launch set that does not capture: say so, keep the row
                tg = float("nan")
                print("   (graph capture failed: %r)" % (ex,))
            print(C, W, dim, "|", B, N, "|", "%-7s" % name, "|", {k: round(v * 1e3, 1) for k, v in p.items()}, "|", round(tot, 1), "|",
                  round(alg / 1e6, 1), "|", round(alg / (tot * 1e-6) / 8e12, 3), "|", round(tg, 1), "|", round(alg / (tg * 1e-6) / 8e12, 3), flush=True)

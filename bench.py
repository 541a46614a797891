#!/usr/bin/env python3
"""Headline benchmark: points/sec of the MHCT hot path (positions -> Splat ->
Slice, forward + backward) on synthetic 4096-point clouds.

  python bench.py --gpus N --steps K --warmup W

A "step" is one fwd+bwd pass of the hot path over one batch of synthetic input
(BASELINE.json north-star op-level shape: B=8, N=4096, H=64 heads, 32x32 grid,
C=16 features/head), inputs resident in HBM.  The step is captured into a HIP
graph so that the timed region is device time, not Python launch time.  One
process per GPU; ranks shard independent clouds (no data-path collective), so
scaling is weak: value = n_gpus * B * N / max-over-ranks step time.

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline     — the dominant kernel's algorithmic bytes / its HIP-event time vs 8 TB/s
  cpu_baseline — the CPU oracle (a port of the reference's PyTorch CPU path)
                 timed on this host on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--points", type=int, default=4096)
    p.add_argument("--heads", type=int, default=64)
    p.add_argument("--feat", type=int, default=16, help="features per head (C)")
    p.add_argument("--grid", type=int, default=32)
    p.add_argument("--dim", type=int, default=2)
    p.add_argument("--reduce", default="max", choices=["max", "sum"])
    p.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    p.add_argument("--graph-steps", type=int, default=10,
                   help="steps captured per HIP graph (the K timed steps are replays of it plus single-step replays for the rest)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg")
    return p.parse_args()


def time_passes(step, iters=30):
    """Average device time (ms) of each of the four ABI passes, measured with HIP
    events on the stream the kernels are launched on (torch's current stream)."""
    res = {}
    for name in step.PASSES:
        fn = getattr(step, name)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters
    return res


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/traffic_latest.json, written by tools/pmc_traffic.py from separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command, with the gfx950
    correction of MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide read)."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    for name, rec in table.get("kernels", {}).items():
        if kernel in name:
            return rec.get("hbm_bytes_per_launch")
    return None


def cpu_baseline(args):
    """The oracle (a pure-PyTorch port of the reference op sequence, materialised
    intermediates included) on a bounded sample: the same workload at batch 1."""
    from oracle import ref_cpu as R
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(1234)
    Bs = 1
    keys = torch.tanh(torch.randn(Bs, args.heads * args.dim, args.points, generator=g))
    feat = torch.randn(Bs, args.heads * args.feat, args.points, generator=g)
    cot = torch.randn(Bs, args.heads * args.feat, args.points, generator=g)
    best = None
    t_start = time.perf_counter()
    reps = 0
    while reps < 5 and (time.perf_counter() - t_start) < args.cpu_seconds:
        t0 = time.perf_counter()
        R.splat_slice_step(keys, feat, cot, args.grid, args.heads, args.dim, args.reduce)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    return {"value": Bs * args.points / best, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": "oracle/ref_cpu.splat_slice_step fwd+bwd, batch %d of the same workload "
                      "(N=%d, H=%d, C=%d, %dD W=%d, reduce=%s), best of %d, torch threads=%d"
                      % (Bs, args.points, args.heads, args.feat, args.dim, args.grid, args.reduce, reps, threads)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from cloud_transformers_amd.step import SplatSliceStep
    from cloud_transformers_amd.parallel import barrier, max_over_ranks

    torch.manual_seed(1234 + rank)
    B, N, H, C, W, dim = args.batch, args.points, args.heads, args.feat, args.grid, args.dim
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    step = SplatSliceStep(keys, feat, cot, W, H, dim, args.reduce)

    # eager warm-up (also sets the LDS attributes before capture)
    step.run()
    torch.cuda.synchronize()
    graph = multi = None
    gs = max(1, min(args.graph_steps, args.steps))
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step.run()
            if gs > 1:                   # `gs` whole steps per graph: one graph launch (and its start-up gap) per gs steps
                multi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(multi):
                    for _ in range(gs):
                        step.run()
                multi.replay()           # set-up, not a counted step: the first launch of a graph uploads it
            graph.replay()
            torch.cuda.synchronize()
        except Exception as ex:          # noqa: BLE001 — capture unavailable: time eager launches instead
            print("bench: HIP graph capture failed (%r), timing eager launches" % (ex,), file=sys.stderr)
            graph = multi = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step.run

    def run_steps(k):
        """exactly k steps"""
        if multi is not None:
            for _ in range(k // gs):
                multi.replay()
            k = k % gs
        for _ in range(k):
            run()

    run_steps(args.warmup)
    barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier(dist)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dist, dt)
    ms = dt / args.steps * 1e3

    if rank == 0:
        alg = step.algorithmic_bytes()
        passes = time_passes(step)
        # dominant KERNEL: the longest pass that is a single launch (slice_bwd is two shorter kernels;
        # profiles/*_kernel_stats.csv lists every kernel's average for cross-checking)
        dom = max(step.SINGLE_KERNEL, key=lambda k: passes[k])
        dom_bytes = alg.get(dom + "_launch", alg[dom])
        achieved = dom_bytes / (passes[dom] * 1e-3) / 1e9
        # the committed PMC passes were collected on the default workload only
        default_shape = (B, N, H, C, W, dim, args.reduce) == (8, 4096, 64, 16, 32, 2, "max")
        traffic = measured_traffic(step.KERNELS.get(dom, dom)) if default_shape else None
        out = {
            "metric": "points/sec fwd+bwd MHCT, 4096-pt batch, 1/2/4/8 MI355X; % HBM roofline",   # BASELINE.json
            "value": world * B * N / (dt / args.steps),
            "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "north-star op-level: B=%d clouds x N=%d pts, H=%d heads, C=%d feat/head, "
                                   "%dD grid W=%d, reduce=%s, keys=tanh(randn), seed 1234+rank"
                                   % (B, N, H, C, dim, W, args.reduce),
                       "per_gpu_batch": B, "parallelism": "replica-sharded clouds x%d (no collective)" % world,
                       "hip_graph": graph is not None, "steps_per_graph": gs if multi is not None else 1},
            "roofline": {"bound": "hbm", "kernel": step.KERNELS.get(dom, dom) if args.reduce == "max" else dom,
                         "pass": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": passes[dom]},
            "passes_ms": passes,
            "step_roofline": {"algorithmic_bytes_per_step": alg["total"],
                              "achieved_GBs": alg["total"] / (ms * 1e-3) / 1e9,
                              "frac_of_8TBs": alg["total"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

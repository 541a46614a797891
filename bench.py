#!/usr/bin/env python3
"""Headline benchmark: points/sec of the MHCT hot path (positions -> Splat ->
Slice, forward + backward) on synthetic 4096-point clouds.

  python bench.py --gpus N --steps K --warmup W [--mode op|ddp-step]

mode op (default).  A "step" is one fwd+bwd pass of the hot path over one batch of synthetic input
(BASELINE.json north-star op-level shape: B=8, N=4096, H=64 heads, 32x32 grid, C=16 features/head), inputs
resident in HBM, the two key cotangents (Slice's and Splat's) summed inside the step as autograd does.  The step
is captured into a HIP graph so that the timed region is device time, not Python launch time.  One process per
GPU; ranks shard independent clouds (no data-path collective), so scaling is weak:
value = n_gpus * B * N / max-over-ranks step time.

mode ddp-step.  The S3DIS-shaped training step (BASELINE configs[2]: stem + 12 MultiHeadUnion blocks + head,
B=8 clouds x 4096 points per GPU, cross-entropy, SGD) under DistributedDataParallel + SyncBatchNorm on RCCL
(train_segmentation.py:58-61,128-130): value = n_gpus * B * N / max-over-ranks step time.

N > 1 without a launcher: this process starts N ranks of itself (cloud_transformers_amd/launch.py) before it
touches the GPU and returns their exit code; under `torch.distributed.run` it is one rank.

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline     — the dominant kernel's algorithmic bytes / its HIP-event time vs 8 TB/s
  cpu_baseline — the CPU oracle (a port of the reference's PyTorch CPU path) timed on this host on a bounded
                 sample of the same workload: all physical cores, and one thread (the reference's own setting,
                 train_segmentation.py:25)
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290
METRIC = "points/sec fwd+bwd MHCT, 4096-pt batch, 1/2/4/8 MI355X; % HBM roofline"   # BASELINE.json


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--mode", default="op", choices=["op", "ddp-step"])
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
    p.add_argument("--seed", type=int, default=1234, help="rank r draws its clouds from seed + r")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-ddp-step", action="store_true", help="N > 1, mode op: do not attach the data-parallel training step")
    p.add_argument("--ddp-steps", type=int, default=10, help="timed steps of the attached data-parallel training step")
    p.add_argument("--ddp-timeout", type=int, default=240, help="seconds the attached step may take before the line goes out without it")
    p.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline leg")
    return p.parse_args()


def time_passes(step, iters=200):
    """Average device time (ms) of each of the four ABI passes, measured with HIP events on the stream the kernels are launched
    on (torch's current stream): each pass `iters` times back to back.  This reads 3-9 % LONGER than the same kernels inside the
    timed (graph-replayed) step — a kernel that follows itself finds its inputs colder than one that follows their producer:
    profiles/r6_headline_replay_trace.txt has the rocprofv3 durations inside the replays (they sum to the step) beside these loops'
    — so the per-pass roofline fractions of the line are conservative.  What was tried instead and reads worse: an event between
    the eager launches of a step (the markers keep consecutive kernels from overlapping their ramps: sum 189.8 us against the
    step's 176.2), graph replays with one pass left out (the differences are marginal costs, not durations: dropping Splat forward
    slows the pass behind it), events recorded inside a captured graph (external events: "disallowed in rocm")."""
    import torch
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


time_passes_back_to_back = time_passes      # (the zoo tools' name for it)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` — its FULL instantiated name as rocprofv3 prints it, template arguments included (a
    substring match would pair any template variant with whatever bytes were last committed: VERDICT r5 #7) — from the COMMITTED
    rocprofv3 PMC passes (profiles/traffic_latest.json, written by tools/pmc_traffic.py from separate --pmc FETCH_SIZE / --pmc
    WRITE_SIZE runs of this same command, with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide
    read).  Not measured in this run; None when the committed record has no kernel of exactly that name."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    rec = table.get("kernels", {}).get(kernel)
    return rec.get("hbm_bytes_per_launch") if rec else None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:      # noqa: BLE001
        pass
    return os.cpu_count() or 1


def _time_oracle(args, batch, threads, budget, min_reps=3, reduce=None):
    """median seconds per fwd+bwd of the oracle at `batch` clouds with `threads` torch threads"""
    import torch
    from oracle import ref_cpu as R
    if reduce is not None:
        args = argparse.Namespace(**dict(vars(args), reduce=reduce))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    keys = torch.tanh(torch.randn(batch, args.heads * args.dim, args.points, generator=g))
    feat = torch.randn(batch, args.heads * args.feat, args.points, generator=g)
    cot = torch.randn(batch, args.heads * args.feat, args.points, generator=g)
    times = []
    t_start = time.perf_counter()
    R.splat_slice_step(keys, feat, cot, args.grid, args.heads, args.dim, args.reduce)      # warm-up (allocator, threads)
    warm = time.perf_counter() - t_start
    while len(times) < min_reps or (len(times) < 5 and time.perf_counter() - t_start + warm < budget):
        t0 = time.perf_counter()
        R.splat_slice_step(keys, feat, cot, args.grid, args.heads, args.dim, args.reduce)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget and len(times) >= min_reps:
            break
    return statistics.median(times), len(times)


def cpu_baseline(args):
    """The oracle (a pure-PyTorch port of the reference op sequence, materialised intermediates included) on a
    bounded sample of the same workload: all physical cores at the full batch, and ONE thread — the setting every
    reference script uses (torch.set_num_threads(1), train_segmentation.py:25) — at batch 1 of it."""
    import torch
    before = torch.get_num_threads()
    cores = physical_cores()
    shape = "N=%d, H=%d, C=%d, %dD W=%d, reduce=%s" % (args.points, args.heads, args.feat, args.dim, args.grid, args.reduce)
    t1, r1 = _time_oracle(args, 1, 1, args.cpu_seconds * 0.3)
    # torch's scatter_reduce / scatter_add_ do not scale with threads (128 cores = one thread): the "all cores"
    # figure is the BEST of a small thread sweep at the full batch, and the sweep is reported
    sweep = {}
    cands = sorted({t for t in (1, 8, 16, 32, 64, cores) if t <= cores})
    for t in cands:
        sweep[t] = _time_oracle(args, args.batch, t, args.cpu_seconds * 0.45 / len(cands), min_reps=1)
    best = min(sweep, key=lambda t: sweep[t][0])
    tn, rn = sweep[best]
    # SURVEY 8(d) names both reductions of the reference's Splat: the other one (max -> torch_scatter's scatter_max, sum ->
    # scatter_add_) on all cores beside the workload's own
    other = "sum" if args.reduce == "max" else "max"
    to, ro = _time_oracle(args, args.batch, best, args.cpu_seconds * 0.25, reduce=other)
    torch.set_num_threads(before)
    other_rec = {"reduce": other, "value": args.batch * args.points / to, "unit": "points/s", "cores": best, "batch": args.batch,
                 "sample": "same workload with reduce=%s (%s), median of %d after a warm-up" %
                           (other, "scatter_add_" if other == "sum" else "scatter_max", ro)}
    return {"other_reduction": other_rec,
            "value": args.batch * args.points / tn, "unit": "points/s", "cores": best, "kind": "port",
            "batch": args.batch, "cpu_model": cpu_model(), "physical_cores": cores,
            "thread_sweep": {str(t): args.batch * args.points / sweep[t][0] for t in cands},
            "sample": "oracle/ref_cpu.splat_slice_step fwd+bwd, the whole workload (batch %d; %s), median of %d after a warm-up, "
                      "best of torch threads in %s: %d (of %d physical cores)" % (args.batch, shape, rn, cands, best, cores),
            "single_thread": {"value": args.points / t1, "unit": "points/s", "cores": 1, "batch": 1,
                              "sample": "same, batch 1 of the workload, torch.set_num_threads(1) as the reference's scripts, "
                                        "median of %d after a warm-up" % r1}}


def count_ties(step):
    """Exact ties of this rank's workload: (cell, channel) pairs of Splat(max) whose maximum is reached by more than one
    contribution bit for bit — each makes the Splat(max) backward redo a four-channel group of its plane (HISTORY.md §5b.1b), so a
    step with ties runs longer than one without.  Counted with torch ops on the GPU, outside the timed region."""
    import torch
    from cloud_transformers_amd import ops
    if step.reduce != "max":
        return 0
    B, H, C, N = step.B, step.H, step.C, step.N
    lc, idx = ops.positions(step.keys, list(step.W), H, step.dim)          # (B, H, V, N)
    V = lc.shape[2]
    ties = 0
    for b in range(B):
        z = step.z[b].reshape(H, C, -1)
        prod = (step.feat[b].reshape(H, C, 1, N) * lc[b].reshape(H, 1, V, N)).reshape(H, C, V * N)
        zc = z.gather(2, idx[b].reshape(H, 1, V * N).expand(H, C, V * N))
        matches = int(((prod == zc) & (zc != 0)).sum())
        ties += matches - int((z != 0).sum())
        del prod, zc
    return ties


def emit(line):
    """Print the result as the LAST line of stdout: RCCL writes its version banner through C stdio, which — when stdout
    is a file or a pipe — sits in libc's buffer until exit and would land after a Python print; flush it first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    sys.stdout.flush()
    print(line, flush=True)


def init_rank():
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPU(s) visible to this rank" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    return rank, local_rank, world, dist


def run_op(args):
    import torch
    rank, local_rank, world, dist = init_rank()
    from cloud_transformers_amd.step import SplatSliceStep
    from cloud_transformers_amd.parallel import barrier, max_over_ranks

    torch.manual_seed(args.seed + rank)
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

    # The per-pass HIP-event timings (the roofline's kernel time) are taken BEFORE the step timing, on every rank: the
    # driver's short runs (W = 5 warm-up steps = 1 ms, K = 20 steps = 4 ms) otherwise measure the GPU's clock ramp after
    # the idle seconds of set-up and graph capture — same build, same box: 150-160 M points/s with W = 5, 167 M with
    # W = 200 (a 20-step burst after idle runs at 0.24 ms per step, tools/dev/lat_probe.py; sustained load: 0.195).  The
    # line says so (`config.order`).
    passes = time_passes(step)
    run_steps(args.warmup)
    barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier(dist)
    dt = time.perf_counter() - t0
    # every rank's own time beside the maximum the value is made of: a rank whose clouds hold an exact tie runs its
    # Splat(max) backward 8-15 us longer (profiles/r5_bench_seeds.txt), and the slowest rank sets the step
    rank_ms = [dt / args.steps * 1e3]
    if dist is not None:
        gathered = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([rank_ms[0]], device="cuda", dtype=torch.float64))
        rank_ms = [float(g.item()) for g in gathered]
    dt = max_over_ranks(dist, dt)
    ms = dt / args.steps * 1e3
    # (counted BEHIND the timed steps: its host synchronisations between the per-pass timings and a short timed region let the
    #  clocks fall again — the driver's `--steps 20 --warmup 5` read 163 M points/s with it in front, 176 M with it here)
    ties = count_ties(step)
    if dist is not None:
        t = torch.tensor([ties], device="cuda", dtype=torch.int64)
        dist.all_reduce(t)
        ties = int(t.item())

    if rank == 0:
        alg = step.algorithmic_bytes()
        tags = step.launch_tags()
        # dominant KERNEL: the longest pass that is a single launch (profiles/*_kernel_stats.csv lists every kernel's
        # average for cross-checking)
        single = [p for p in step.PASSES if "+" not in tags[p]] or list(step.PASSES)
        dom = max(single, key=lambda k: passes[k])
        dom_bytes = alg[dom]
        achieved = dom_bytes / (passes[dom] * 1e-3) / 1e9
        # the committed PMC passes were collected on the default workload only
        default_shape = (B, N, H, C, W, dim, args.reduce) == (8, 4096, 64, 16, 32, 2, "max")
        exact = step.HEADLINE_KERNELS if default_shape else {}      # full instantiated names: only the default workload has a PMC record
        traffic = measured_traffic(exact[dom]) if default_shape else None
        roofline_passes = {}
        for p in step.PASSES:
            gbs = alg[p] / (passes[p] * 1e-3) / 1e9
            kern = step.KERNEL_OF.get(tags[p], tags[p])
            roofline_passes[p] = {"kernel": kern, "bytes": alg[p], "ms": passes[p], "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                                  "traffic": measured_traffic(exact[p]) if default_shape and "+" not in tags[p] else None}
        out = {
            "metric": METRIC,
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
                                   "%dD grid W=%d, reduce=%s, keys=tanh(randn), seed %d+rank; step = Splat fwd, Slice fwd, "
                                   "Slice bwd, Splat bwd (key cotangents summed)"
                                   % (B, N, H, C, dim, W, args.reduce, args.seed),
                       "per_gpu_batch": B, "parallelism": "replica-sharded clouds x%d (no collective)" % world,
                       "world_size_seen": world,
                       "hip_graph": graph is not None, "steps_per_graph": gs if multi is not None else 1,
                       "order": "per-pass HIP-event timings (203 launches of each pass back to back: 3-9 % longer than inside the "
                                "replayed step, profiles/r6_headline_replay_trace.txt), then W warm-up steps, then the K timed steps"},
            "roofline": {"bound": "hbm", "kernel": step.KERNEL_OF.get(tags[dom], tags[dom]),
                         "pass": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "committed PMC pass (profiles/traffic_latest.json), not measured in this run"
                                           if traffic is not None else None,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": passes[dom]},
            # every pass of the step against the same roofline, and the one furthest below it: the headline `roofline`
            # object is the LONGEST kernel, which need not be the least efficient one
            "roofline_passes": roofline_passes,
            "worst_pass": min(roofline_passes, key=lambda k: roofline_passes[k]["frac"]),
            "passes_ms": passes,
            # exact ties of the Splat(max) maxima in the clouds of ALL ranks (each costs its plane's workgroup a redo: the
            # step of a rank with ties runs a few per cent longer, and the value is the max over ranks)
            "ties_seen": ties,
            "rank_ms_per_step": rank_ms,
            "kernels": {p: step.KERNEL_OF.get(t, t) for p, t in tags.items()},
            "step_roofline": {"algorithmic_bytes_per_step": alg["total"],
                              "achieved_GBs": alg["total"] / (ms * 1e-3) / 1e9,
                              "frac_of_8TBs": alg["total"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    else:
        out = {}
    if dist is not None and not args.no_ddp_step:
        # N > 1 ranks: the op-level numbers above are taken; now the SAME ranks run the data-parallel training step (mode ddp-step's
        # workload: DDP gradient all-reduce + SyncBatchNorm exchanges on RCCL) and rank 0 attaches it to the ONE line — `value`
        # stays the op-level throughput, `ddp_step.value` is what BASELINE's ">= 6x DDP throughput at 8 GPUs vs 1" is read from
        # (the 1-GPU denominator: `bench.py --mode ddp-step`, profiles/*_bench_line_ddp_step.json)
        del step, keys, feat, cot
        torch.cuda.empty_cache()
        guard = ddp_step_deadline(out, rank, args.ddp_timeout)       # a hung collective must not cost the op-level line
        attach_ddp_step(out, lambda: ddp_step_measure(dist, rank, world, args.ddp_steps, 3, args.batch, args.points, local_rank,
                                                      no_graph=args.no_graph))
        guard.cancel()
    if rank == 0:
        if dist is not None:
            dist.barrier()             # every rank's banners are out (they flush below before this barrier completes)
        emit(json.dumps(out))
    elif dist is not None:
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def ddp_step_measure(dist, rank, world, steps, warmup, batch, points, device_index, no_graph=False, make=None):
    """The S3DIS-shaped segmenter training step under DDP + SyncBatchNorm, timed in the calling rank of an initialised process
    group (train_segmentation.py:128-130: SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel; gradient all-reduce
    as utils/train_util_distributed.py:12-34 averages them).  Every rank returns the same-shaped dict:
      ms_per_step, value (world * batch * points / s), collectives_per_step (norm-statistics exchanges), params_equal_across_ranks,
      param_checksum_spread, step ("graph" | "eager" | "eager (graph capture failed on some rank: ...)"), loss, parameters.
    Forward + loss + backward run as ONE HIP graph with the RCCL collectives captured; the ranks AGREE on graph-or-eager (an
    all-reduced flag after the capture: a rank replaying a graph beside ranks stepping eagerly would deadlock in DDP's
    reducer), and the eager fallback starts from a FRESH DDP wrapper and optimizer (a capture that failed inside backward leaves
    the old reducer mid-iteration).  `make` (tests: a CPU stand-in under gloo) -> (net, input, labels) on the target device;
    `device_index` None = CPU (no graphs, no streams)."""
    import torch
    from torch import nn
    from cloud_transformers_amd.parallel import barrier, data_parallel, max_over_ranks, quiesce as parallel_quiesce
    from cloud_transformers_amd import ops
    on_gpu = device_index is not None
    B, N = batch, points
    torch.manual_seed(0)                               # same initial weights on every rank
    if make is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from segmenter_step_bench import Segmenter
        net = Segmenter().cuda()
    prev_stream = work_stream = None
    if on_gpu:
        # Everything DDP runs on ONE side stream — its construction (the reducer stashes the parameters' AccumulateGrad nodes,
        # which remember the stream they were made on: its gradient hooks, and with them the bucketed all-reduce, run THERE),
        # the warm-up iterations, the capture and the replays: torch's recipe for capturing DDP (notes/cuda.rst).
        prev_stream = torch.cuda.current_stream()
        work_stream = torch.cuda.Stream()
        work_stream.wait_stream(prev_stream)
        torch.cuda.set_stream(work_stream)
    try:
        torch.manual_seed(1234 + rank)                     # its own shard of the batch
        if make is None:
            cloud = torch.cat([torch.rand(B, 3, N, device="cuda") * 2 - 1, torch.rand(B, 3, N, device="cuda")], dim=1)
            labels = torch.randint(13, (B, N), device="cuda")
        else:
            torch.manual_seed(0)
            net, cloud, labels = make(rank)
        state = {}

        def wrap():
            state["ddp"] = data_parallel(net, device_index)
            state["opt"] = torch.optim.SGD(state["ddp"].parameters(), lr=0.01, momentum=0.9)

        wrap()
        lossf = nn.CrossEntropyLoss()

        def fwd_bwd():
            loss = lossf(state["ddp"](cloud), labels)
            loss.backward()                                # bucketed gradient all-reduce overlaps with this
            return loss

        def one_eager():
            state["opt"].zero_grad(set_to_none=True)
            loss = fwd_bwd()
            state["opt"].step()
            return loss

        # DDP wants 11 eager iterations before a capture (its reducer rebuilds the buckets after the first and settles)
        for _ in range(max(11, warmup) if on_gpu else max(2, warmup)):
            one_eager()
        graph, graph_error, collectives, static_loss = None, None, None, None
        if on_gpu and not no_graph:
            torch.cuda.synchronize()
            barrier(dist)
            parallel_quiesce()
            try:
                state["opt"].zero_grad(set_to_none=True)
                c0 = ops.sync_stats_collectives()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=work_stream, capture_error_mode="thread_local"):
                    static_loss = fwd_bwd()
                collectives = ops.sync_stats_collectives() - c0        # enqueued once, at capture; replayed every step
            except Exception as e:      # noqa: BLE001
                graph, graph_error = None, "%s: %s" % (type(e).__name__, str(e)[:200])
                torch.cuda.synchronize()
            ok = torch.tensor([1 if graph is not None else 0], device="cuda", dtype=torch.int32)
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if graph is not None:
                    graph_error = "another rank's capture failed"
                graph, collectives = None, None
                torch.cuda.synchronize()
                wrap()                                     # fresh reducer, fresh optimizer state: the eager step in a clean state
                for _ in range(3):
                    one_eager()

        def one():
            if graph is None:
                return one_eager()
            graph.replay()
            state["opt"].step()
            return static_loss

        for _ in range(3 if on_gpu else 1):
            one()
        coll0 = ops.sync_stats_collectives()
        barrier(dist)
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        if on_gpu:
            torch.cuda.synchronize()
        barrier(dist)
        dt = max_over_ranks(dist, time.perf_counter() - t0)
        if collectives is None:
            collectives = (ops.sync_stats_collectives() - coll0) / steps
        # every rank must hold the same parameters and running statistics after the steps (same initial weights, averaged
        # gradients, job-wide batch statistics): one float64 checksum per rank, gathered
        with torch.no_grad():
            chk = torch.stack([t.double().sum() for t in list(net.parameters()) + [b for b in net.buffers() if b.is_floating_point()]]).sum().reshape(1)
        chks = [torch.zeros_like(chk) for _ in range(world)]
        if world > 1:
            dist.all_gather(chks, chk)
        else:
            chks = [chk]
        spread = float((torch.stack(chks) - chks[0]).abs().max() / chks[0].abs().clamp_min(1e-30))
        return {
            "ms_per_step": dt / steps * 1e3, "value": world * B * N / (dt / steps), "unit": "points/s", "steps": steps,
            "collectives_per_step": collectives, "params_equal_across_ranks": spread <= 1e-9, "param_checksum_spread": spread,
            "step": "graph" if graph is not None else ("eager" + (" (graph capture failed: %s)" % graph_error if graph_error else "")),
            "loss": float(loss.detach()), "parameters": sum(p.numel() for p in net.parameters()),
            "per_gpu_batch": B, "points": N, "world_size_seen": world,
        }
    finally:
        if on_gpu:
            torch.cuda.synchronize()
            torch.cuda.set_stream(prev_stream)


def ddp_step_deadline(out, rank, seconds):
    """A timer per rank: if the attached data-parallel step has not come back after `seconds` (a collective that never completes
    cannot be cancelled from Python), rank 0 prints the op-level line with ddp_step.error set and every rank leaves through
    os._exit(0) — the line the driver came for is not lost to the attachment."""
    import threading

    def fire():
        if rank == 0:
            out["ddp_step"] = {"error": "no result after %d s (a collective did not complete?); op-level numbers above are unaffected" % seconds}
            emit(json.dumps(out))
        os._exit(0)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def attach_ddp_step(out, measure):
    """out["ddp_step"] = measure(), or {"error": ...}: the op-level line is never lost to a failure of the attached leg (every rank
    calls this; rank 0's `out` is the one printed, the others pass a scratch dict)."""
    try:
        out["ddp_step"] = measure()
    except Exception as e:      # noqa: BLE001
        out["ddp_step"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return out


def run_ddp_step(args):
    """S3DIS-shaped segmenter training step under DDP + SyncBatchNorm (also at world size 1: the wrap is the same)."""
    import torch
    rank, local_rank, world, dist = init_rank()
    import torch.distributed as tdist
    if world == 1:
        # one rank: keep the norms' statistics exchange ON (a one-rank RCCL group), so that the step enqueues — and this
        # line times — what every rank of an N-GPU step enqueues: 76 collectives + the bucketed gradient all-reduce
        os.environ.setdefault("CLOUDCT_SYNCBN_FORCE", "1")
    if dist is None:            # DDP needs a process group even for one rank
        from cloud_transformers_amd.launch import free_port
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        dist = tdist
    from cloud_transformers_amd import ops
    B, N = args.batch, args.points
    m = ddp_step_measure(dist, rank, world, args.steps, args.warmup, B, N, local_rank, no_graph=args.no_graph)
    if rank == 0:
        nbytes = m["parameters"] * 4
        out = {
            "metric": METRIC, "value": m["value"], "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S3DIS-shaped segmenter training step (stem + 12 MultiHeadUnion + head, %.1f M parameters), "
                                   "B=%d clouds x N=%d pts per GPU, cross-entropy + SGD, DistributedDataParallel + SyncBatchNorm on RCCL"
                                   % (nbytes / 4e6, B, N),
                       "per_gpu_batch": B, "parallelism": "dp%d" % world, "world_size_seen": world,
                       "gradient_allreduce_MB_per_step": nbytes / 1e6,
                       "norm_statistics_collectives_per_step": m["collectives_per_step"],
                       "norm_statistics_exchange": "forced on at world size 1 (CLOUDCT_SYNCBN_FORCE)" if (world == 1 and ops.SYNC_STATS_FORCE)
                                                   else ("on" if world > 1 else "off (one rank: plain batch norm)"),
                       "step": "HIP graph (forward + loss + backward, collectives captured) + optimizer" if m["step"] == "graph"
                               else m["step"],
                       "params_equal_across_ranks": m["params_equal_across_ranks"], "param_checksum_spread": m["param_checksum_spread"],
                       "loss": m["loss"]},
        }
        dist.barrier()
        emit(json.dumps(out))
    else:
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    from cloud_transformers_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # parent: start one rank per GPU before anything here touches the GPU, return their exit code.  The library is
        # built HERE, once (compile only, no GPU call): N ranks finding it missing would all run hipcc at the same time.
        from cloud_transformers_amd import _lib
        _lib.build()
        have = launch.visible_gpus()                   # KFD topology + *_VISIBLE_DEVICES: no HIP runtime in the parent
        if have is not None and have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
        raise SystemExit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                                            capture=not args.no_graph and (args.mode == "ddp-step" or not args.no_ddp_step)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launch.under_launcher() and world != args.gpus:
        print("bench: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d" % (args.gpus, world, world),
              file=sys.stderr)
    captures_collectives = not args.no_graph and (args.mode == "ddp-step" or (world > 1 and not args.no_ddp_step))
    if captures_collectives:
        # torch's recipe for capturing collectives (notes/cuda.rst): with the process group's watchdog polling events of a
        # capturing stream the capture dies.  The price — a hung collective is not aborted, the job hangs instead of exiting —
        # is paid only by runs that capture (launch.rank_env(capture=True) does the same for spawned ranks)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    if args.mode == "ddp-step":
        if args.steps == 200 and args.warmup == 20:      # defaults are the op benchmark's: a training step is ~100x longer
            args.steps, args.warmup = 20, 3
        return run_ddp_step(args)
    return run_op(args)


if __name__ == "__main__":
    main()

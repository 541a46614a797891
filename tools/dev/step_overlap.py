"""Headline step as a HIP graph: in-kernel sort vs record sorted on the main stream vs record sorted on a side stream beside the forward passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import _lib
from cloud_transformers_amd.step import SplatSliceStep

lib = _lib.load()
B, N, H, C, W, dim = 8, 4096, 64, 16, 32, 2
torch.manual_seed(1234)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")


def timed_graph(fn, reps=200):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


step = SplatSliceStep(keys, feat, cot, W, H, dim, "max", plane_sort=True)
rec = step.sorted
side = torch.cuda.Stream()


def run_scatter():
    step.splat_fwd(); step.slice_fwd(); step.slice_bwd(); step.splat_bwd()


def run_inkernel():
    step.sorted = None
    run_scatter()


def run_record():
    step.sorted = rec
    step.plane_sort(); run_scatter()


def run_record_side():
    step.sorted = rec
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        step.plane_sort()
    step.splat_fwd(); step.slice_fwd()
    cur.wait_stream(side)
    step.slice_bwd(); step.splat_bwd()


lib.ct_debug_set_flags(_lib.DEBUG_NO_SORTED)
print("scatter form           : %.1f us" % timed_graph(run_scatter), flush=True)
lib.ct_debug_set_flags(0)
print("sorted, sort inside    : %.1f us" % timed_graph(run_inkernel), flush=True)
print("sorted, record         : %.1f us" % timed_graph(run_record), flush=True)
print("sorted, record on side : %.1f us" % timed_graph(run_record_side), flush=True)

"""The two native losses beside THE REFERENCE'S OWN KERNELS on the same MI355X: chamfer_extension/chamfer.cu and
emd_linear/emd_cuda.cu compiled for gfx950 (oracle/_ref/chamfer_reference.so, emd_reference.so: `make -C oracle ref`, test
infrastructure) against ct_chamfer_fwd / ct_emd_fwd, forward only, device time by HIP events, on the completion and reconstruction
shapes — CUDA kernels recompiled for CDNA4 are the baseline a port would deliver."""
import importlib.util
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloud_transformers_amd.chamfer import chamfer_with_indices
from cloud_transformers_amd.emd import emdModule

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def timeit(f, iters):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cham, emd_ref = load("chamfer_reference"), load("emd_reference")
for B, n in [(2, 16384), (4, 8192), (8, 2048)]:
    torch.manual_seed(0)
    a, b = torch.rand(B, n, 3, device="cuda"), torch.rand(B, n, 3, device="cuda")
    d1, d2 = torch.zeros(B, n, device="cuda"), torch.zeros(B, n, device="cuda")
    i1, i2 = torch.zeros(B, n, dtype=torch.int32, device="cuda"), torch.zeros(B, n, dtype=torch.int32, device="cuda")
    t_ref = timeit(lambda: cham.forward(a, b, d1, d2, i1, i2), 10)
    t_mine = timeit(lambda: chamfer_with_indices(a, b), 10)
    print("chamfer fwd B%d n=m=%d: reference kernels %.1f us | ct_chamfer_fwd %.1f us | x%.1f" % (B, n, t_ref * 1e3, t_mine * 1e3, t_ref / t_mine))
    i32 = dict(dtype=torch.int32, device="cuda")

    def ref_emd():
        bufs = (torch.zeros(B, n, device="cuda"), torch.zeros(B, n, **i32) - 1, torch.zeros(B, n, device="cuda"), torch.zeros(B, n, **i32) - 1,
                torch.zeros(B, n, **i32), torch.zeros(B, n, device="cuda"), torch.zeros(B, n, device="cuda"), torch.zeros(B * n, **i32),
                torch.zeros(512, **i32), torch.zeros(512, **i32), torch.zeros(512, **i32), torch.zeros(B * n, **i32))
        dist, ass, price, ass_inv, bid, bid_inc, max_inc, unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx = bufs
        emd_ref.forward(a, b, dist, ass, price, ass_inv, bid, bid_inc, max_inc, unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, 0.005, 50)
    m = emdModule()
    t_ref = timeit(ref_emd, 3)
    t_mine = timeit(lambda: m(a, b, 0.005, 50), 3)
    print("emd fwd     B%d n=%d (eps 0.005, 50 iterations): reference kernels %.2f ms | ct_emd_fwd %.2f ms | x%.1f" % (B, n, t_ref, t_mine, t_ref / t_mine))

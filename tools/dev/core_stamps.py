"""Phase timeline of the plane-resident core from a CT_CORE_STAMP build (tools/dev/build_core_abl.sh stamp -DCT_CORE_STAMP=1;
CLOUDCT_LIB=.../libcloudct_stamp.so python tools/dev/core_stamps.py): median over workgroups of the time between the stamps
(s_memrealtime, 10 ns ticks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream

NAMES = ["zero", "scatter", "exchange", "side outputs", "conv 0", "gather 0", "conv 1", "gather 1"]
lib = _lib.load()
for B, H, N, dim, W, C in [(8, 64, 4096, 2, 32, 16), (8, 16, 4096, 2, 16, 16), (8, 16, 4096, 3, 8, 32), (2, 16, 16384, 3, 8, 32)]:
    torch.manual_seed(0)
    Wl = [W] * dim
    Wa = _lib.int_array(Wl)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    w = torch.randn(H * C, C, *([3] * dim), device="cuda") / (C * 3 ** dim) ** 0.5
    bias = torch.randn(H * C, device="cuda") * 0.1
    out = torch.empty_like(feat)
    z = torch.empty(B, H * C, *Wl, device="cuda"); y = torch.empty_like(z)
    occ = torch.empty((), device="cuda", dtype=torch.int64)
    nws = lib.ct_mhct_core_workspace_bytes(B, H, C, N, dim, Wa)
    ws = torch.zeros(nws, device="cuda", dtype=torch.uint8)
    for _ in range(20):
        _lib.check(lib.ct_mhct_core_fwd(_ptr(keys), _ptr(feat), None, 0, _ptr(w), _ptr(bias), _ptr(out), _ptr(z), _ptr(y), _ptr(occ),
                                        _ptr(ws), nws, B, H, C, N, dim, Wa, _stream()), "core")
    torch.cuda.synchronize()
    # the stamps sit at the end of the workspace: [workgroups][16] u64
    tail = ws.cpu().numpy()
    # recover the layout from the size: stamps are the last nwg*128 bytes for the S the planner chose
    for S in (8, 4, 2, 1):
        n = B * H * S
        if n * 128 > nws:
            continue
        st = np.frombuffer(tail[nws - n * 128:].tobytes(), dtype=np.uint64).reshape(n, 16)
        if st[:, 0].min() > 0 and (st[:, 1] >= st[:, 0]).all():
            break
    d = np.diff(st[:, :9].astype(np.int64), axis=1) * 0.01      # us
    nseg = 6 if (C // 16) // (2 if (C == 32 and S >= 2) else 1) == 1 else 8
    span = (st[:, :nseg + 1].max() - st[:, 0].min()) * 0.01
    print(f"B{B} H{H} N{N} {dim}D W{W} C{C}  S={S}: kernel span {span:.1f} us; first-to-last start {(st[:, 0].max() - st[:, 0].min()) * 0.01:.1f} us")
    for k in range(nseg):
        print(f"    {NAMES[k]:14s} median {np.median(d[:, k]):6.1f} us   p90 {np.percentile(d[:, k], 90):6.1f}")

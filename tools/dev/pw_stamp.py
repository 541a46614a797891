"""Phase stamps of ct_pw_gemm's K-step (development build: bash tools/dev/build_exp.sh 50 -DPW_STAMP;
CLOUDCT_LIB=.../libcloudct_exp50.so python tools/dev/pw_stamp.py [B Co Ci N]): mean cycles per K-step and wave spent in
loop top -> fetch of step kt+2 issued -> (the loads of step kt+1 have landed: explicit vmcnt wait) -> fragment reads, MFMAs
of step kt with the split + LDS stores of step kt+1 between them -> barrier passed (s_memtime; each stamp drains the LDS
queue first)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import _lib, ops

B, Co, Ci, N = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 848, 512, 4096)
lib = ctypes.CDLL(_lib.LIB_PATH)
torch.manual_seed(0)
W = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
x = torch.randn(B, Ci, N, device="cuda")
gy = torch.randn(B, Co, N, device="cuda")
am_w, am_x, am_g = ops.amax(W), ops.amax(x), ops.amax(gy)
buf = torch.zeros(4096 * 8 * 6, dtype=torch.int64, device="cuda")
lib.ct_debug_pw_stamp.argtypes = [ctypes.c_void_p]
for mode, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
    args = {0: (W, x, am_w, am_x), 1: (W, gy, am_w, am_g), 2: (gy, x, am_g, am_x)}[mode]
    lib.ct_debug_pw_stamp(None)
    for _ in range(3):
        ops.pw_gemm(mode, *args, B, Co, Ci, N)
    buf.zero_()
    torch.cuda.synchronize()
    lib.ct_debug_pw_stamp(ctypes.c_void_p(buf.data_ptr()))
    ops.pw_gemm(mode, *args, B, Co, Ci, N)
    torch.cuda.synchronize()
    t = buf.view(4096, 8, 6).double()
    used = t.sum(dim=(1, 2)) > 0
    t = t[used]
    steps = (Ci if mode == 0 else Co if mode == 1 else None)
    tot = t.sum(dim=2).mean()
    frac = t.mean(dim=(0, 1)) / tot
    print(name, "blocks", int(used.sum()), "cycles per wave in the loop %.0f" % float(tot),
          "| share: fetch issue %.2f  waiting for loads %.2f  MFMAs + split + LDS stores %.2f  barrier %.2f  loop top %.2f" %
          (float(frac[1]), float(frac[2] + frac[3]), float(frac[4]), float(frac[5]), float(frac[0])))

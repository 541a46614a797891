"""Phase stamps of pw2_gemm_kernel's K-step (development build: bash tools/dev/build_exp.sh 51 -DPW_STAMP;
CLOUDCT_LIB=.../libcloudct_exp51.so python tools/dev/pw2_stamp.py [B Co Ci N]): mean cycles per step and wave between the
five s_memtime stamps of a step, the time before the loop and the whole kernel."""
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
buf = torch.zeros(512 * 8 * 8, dtype=torch.int64, device="cuda")
lib.ct_debug_pw_stamp.argtypes = [ctypes.c_void_p]
for mode, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
    args = {0: (W, x, am_w, am_x), 1: (W, gy, am_w, am_g), 2: (gy, x, am_g, am_x)}[mode]
    lib.ct_debug_pw_stamp(None)
    for _ in range(20):
        ops.pw_gemm(mode, *args, B, Co, Ci, N)
    buf.zero_()
    torch.cuda.synchronize()
    lib.ct_debug_pw_stamp(ctypes.c_void_p(buf.data_ptr()))
    ops.pw_gemm(mode, *args, B, Co, Ci, N)
    torch.cuda.synchronize()
    t = buf.view(512, 8, 8).double()
    used = t[:, :, 7].sum(dim=1) > 0
    t = t[used]
    steps = t[:, :, 5].mean()
    per = t[:, :, :5].sum(dim=(0, 1)) / t[:, :, 5].sum()
    print("%s: workgroups %d  steps per wave %.1f (min %d max %d)  cycles per step %.0f = ks0 groups %.0f | ks1 groups 1-2 %.0f | loads, "
          "fragments, last group %.0f | barrier %.0f | to the next top %.0f ;  before the loop %.0f  whole kernel %.0f (max %.0f)  loop share %.2f"
          % (name, int(used.sum()), float(steps), int(t[:, :, 5].min()), int(t[:, :, 5].max()), float(per.sum()), *[float(v) for v in per],
             float(t[:, :, 6].mean()), float(t[:, :, 7].mean()), float(t[:, :, 7].max()),
             float(t[:, :, :5].sum() / t[:, :, 7].sum())))

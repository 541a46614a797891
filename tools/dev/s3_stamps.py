"""tools/dev/s3_stamps.py [C W B N]: the sorted 3D Slice backward's phase stamps (library built with -DCT_SORT_STAMPS: workgroup
(0,0,0)'s clock at the phase boundaries) and its time, on one zoo head shape (default 32 8 8 4096)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.step import SplatSliceStep
C, W, B, N = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (32, 8, 8, 4096)
H, dim = 16, 3
lib = _lib.load()
torch.manual_seed(0)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
for _ in range(50):
    st.run()
torch.cuda.synchronize()
print(st.launch_tags())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    st.slice_bwd()
e1.record()
torch.cuda.synchronize()
print("slice_bwd %.1f us" % (e0.elapsed_time(e1) * 10))
if hasattr(lib, "ct_debug_sorted_stamps"):
    buf = (ctypes.c_ulonglong * 64)()
    lib.ct_debug_sorted_stamps.argtypes = [ctypes.c_void_p]
    st.slice_bwd(); torch.cuda.synchronize()
    lib.ct_debug_sorted_stamps(buf)
    t = [buf[i] for i in range(64)]
    print("kernel (WG 0): sort %d  weights+barrier %d  groups %d  epilogue %d  fold %d   total %d" % (
        t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
    gn = ["wait+max", "stage", "barrier1", "-", "item0", "item1", "barrier2", "writeout"]
    print("second group:", " ".join("%s:%d" % (gn[i], t[17 + i] - t[16 + i]) for i in range(7)))
    t0 = min(t[24 + w] for w in range(8))
    print("second group items per wave (start .. end):", " ".join("w%d:%d..%d" % (w, t[24 + w] - t0, t[40 + w] - t0) for w in range(8)))
if hasattr(lib, "ct_debug_wg_stamps"):
    import numpy as np
    wb = (ctypes.c_ulonglong * (4096 * 4))()
    lib.ct_debug_wg_stamps.argtypes = [ctypes.c_void_p]
    lib.ct_debug_wg_stamps(wb)
    a = np.frombuffer(wb, dtype=np.uint64).reshape(4096, 4)
    a = a[a[:, 1] > 0]
    t0 = int(a[:, 0].min())
    start, end = a[:, 0].astype(np.int64) - t0, a[:, 1].astype(np.int64) - t0
    hw, xcc = a[:, 2].astype(np.int64), a[:, 3].astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 0x1
    print("workgroups %d: duration min/median/max %d/%d/%d cycles" % (len(a), (end - start).min(), np.median(end - start), (end - start).max()))
    place = xcc * 1000 + se * 100 + sh * 50 + cu
    uniq, cnts = np.unique(place, return_counts=True)
    print("distinct (xcc, se, sh, cu): %d; workgroups per CU min/max %d/%d; per XCC %s" % (len(uniq), cnts.min(), cnts.max(), np.bincount(xcc).tolist()))
    # (entry / exit clocks are per XCD — s_memtime is not synchronised across them — so only DURATIONS are compared)

"""Which tie path a Splat(max) backward launch took (library built with -DCT_TIE_DEBUG; CLOUDCT_LIB=...):
python tools/dev/tie_debug.py C W dim B N H seed"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.step import SplatSliceStep
C, W, dim, B, N, H, seed = [int(v) for v in sys.argv[1:8]]
lib = _lib.load()
fn = lib.ct_debug_tie_counters
torch.manual_seed(seed)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
st = SplatSliceStep(keys, feat, cot, W, H, dim, "max")
buf = (ctypes.c_uint * 16)()
st.run(); torch.cuda.synchronize()
fn(buf)
st.run(); torch.cuda.synchronize()
fn(buf)
names = ["searches", "candidates", "resident", "repaired", "failed tries", "group redos", "plane redos", "tied groups", "surplus",
         "repaired through memory", "memory repairs failed"]
print("seed", seed, {n: buf[i] for i, n in enumerate(names)})

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.ops import _ptr, _stream
lib = _lib.load()
B, G, Ci, Co, W = 8, 16, 64, 64, (8, 8, 8)
x = torch.randn(B, G * Ci, *W, device="cuda"); w = torch.randn(G * Co, Ci, 3, 3, 3, device="cuda") * 0.05
y = torch.empty(B, G * Co, *W, device="cuda"); Wa = _lib.int_array(W)
for _ in range(5):
    _lib.check(lib.ct_gconv_fwd(_ptr(x), _ptr(w), None, _ptr(y), B, G, Ci, Co, 3, Wa, _stream()), "f")
torch.cuda.synchronize()

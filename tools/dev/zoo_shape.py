"""tools/dev/zoo_shape.py C W dim B N [reduce]: launch tags and per-pass times of ONE zoo head shape (H16) — the quick A/B probe
(environment switches: CLOUDCT_WIDE, CLOUDCT_SORTED, CLOUDCT_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
from bench import time_passes_back_to_back as time_passes
C, W, dim, B, N = [int(v) for v in sys.argv[1:6]]
reduce = sys.argv[6] if len(sys.argv) > 6 else "max"
H = 16
torch.manual_seed(0)
keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
feat = torch.randn(B, H * C, N, device="cuda")
cot = torch.randn(B, H * C, N, device="cuda")
if os.environ.get("CT_FLAGS"):        # ct_debug_set_flags bits (include/cloudct.h: CT_DEBUG_*), e.g. 2 = FORCE_HOT
    from cloud_transformers_amd import _lib
    _lib.load().ct_debug_set_flags(int(os.environ["CT_FLAGS"]))
st = SplatSliceStep(keys, feat, cot, W, H, dim, reduce)
tags = st.launch_tags()
for _ in range(100):
    st.run()
torch.cuda.synchronize()
p = time_passes(st, iters=100)
tot = sum(p.values()) * 1e3
alg = st.algorithmic_bytes()["total"]
print(C, W, dim, "|", B, N, "|", {k: round(v * 1e3, 1) for k, v in p.items()}, "|", round(tot, 1), "|", round(alg / (tot * 1e-6) / 8e12, 3),
      "| WIDE=%s SORTED=%s FLAGS=%s NSEG=%s" % (os.environ.get("CLOUDCT_WIDE", "-"), os.environ.get("CLOUDCT_SORTED", "-"),
                                                 os.environ.get("CT_FLAGS", "-"), os.environ.get("CLOUDCT_SPLAT_BWD_NSEG", "-")))
print("   ", tags)

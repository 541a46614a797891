import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd import _lib
from cloud_transformers_amd.step import SplatSliceStep
lib = _lib.load()
B, N, H, C, W, dim = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (8, 4096, 16, 16, 16, 2)
dup = len(sys.argv) > 7 and sys.argv[7] == "dup"
for seed in range(100, 106):
    g = torch.Generator(device="cuda").manual_seed(seed)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda", generator=g))
    feat = torch.randn(B, H * C, N, device="cuda", generator=g)
    cot = torch.randn(B, H * C, N, device="cuda", generator=g)
    if dup:
        keys[..., N // 2:] = keys[..., :N // 2]; feat[..., N // 2:] = feat[..., :N // 2]
    res = {}
    for name, flags in (("hot", 0), ("generic", _lib.DEBUG_NO_HOT if hasattr(_lib, "DEBUG_NO_HOT") else 1)):
        lib.ct_debug_set_flags(flags)
        st = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=False)
        st.run(); torch.cuda.synchronize()
        res[name] = (st.g_feat.clone(), st.g_keys_out.clone(), st.launch_tags()["splat_bwd"])
        lib.ct_debug_set_flags(0)
    gf = (res["hot"][0] - res["generic"][0]).abs().max().item()
    gk = (res["hot"][1] - res["generic"][1]).abs().max().item()
    if dup:
        h = N // 2
        gk = ((res["hot"][1][..., :h] + res["hot"][1][..., h:]) - (res["generic"][1][..., :h] + res["generic"][1][..., h:])).abs().max().item()
        gf = ((res["hot"][0][..., :h] + res["hot"][0][..., h:]) - (res["generic"][0][..., :h] + res["generic"][0][..., h:])).abs().max().item()
    print(seed, res["hot"][2], res["generic"][2], "max |dg_feat| %.3g  max |dg_keys| %.3g  (max |g_keys| %.3g)" % (gf, gk, res["generic"][1].abs().max().item()))

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloud_transformers_amd.step import SplatSliceStep
for shape in [(8, 4096, 16, 16, 16, 2), (1, 8192, 2, 4, 32, 2), (8, 4096, 16, 16, 16, 3)]:
    B, N, H, C, W, dim = shape
    torch.manual_seed(0)
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda"))
    feat = torch.randn(B, H * C, N, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda")
    a = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=True)
    b = SplatSliceStep(keys, feat, cot, W, H, dim, "max", tickets=False)
    if B * H < 32:
        a.lib.ct_debug_set_flags(2)
    for st in (a, b):
        st.splat_fwd(); st.slice_fwd(); st.slice_bwd()
    torch.cuda.synchronize()
    print(shape, "tags", a.lib.ct_debug_last_launch().decode())
    for name in ("g_z", "g_keys_buf"):
        x, y = getattr(a, name), getattr(b, name)
        d = (x - y).abs()
        print(" slice_bwd", name, "equal", torch.equal(x, y), "max", float(d.max()))
        if not torch.equal(x, y):
            bad = (d > 0).reshape(B, -1, *x.shape[2:])
            rows = bad.flatten(2).any(-1)
            print("  bad rows per cloud:", rows.sum(1).tolist(), "of", rows.shape[1])
            r = bad.flatten(2)[rows][0]
            idx = r.nonzero().flatten()
            print("  first bad row: n range", int(idx.min()), int(idx.max()), "count", int(idx.numel()))
    for st in (a, b):
        st.splat_bwd()
    torch.cuda.synchronize()
    print(" tags", a.lib.ct_debug_last_launch().decode())
    for name in ("g_feat", "g_keys_out"):
        x, y = getattr(a, name), getattr(b, name)
        print(" splat_bwd", name, "equal", torch.equal(x, y), "max", float((x - y).abs().max()))
    print(" tickets nonzero:", int((a.tickets != 0).sum()))
    a.lib.ct_debug_set_flags(0)

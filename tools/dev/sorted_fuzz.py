"""Random layouts: sorted-plane Slice backward / Splat(sum) forward (forced) against the scatter form on the same inputs.
python tools/dev/sorted_fuzz.py [cases] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd import _lib, ops
from cloud_transformers_amd.ops import _ptr, _stream

lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def per_channel_err(a, b):
    a, b = a.double(), b.double()
    B, HC = a.shape[:2]
    a, b = a.reshape(B, HC, -1), b.reshape(B, HC, -1)
    return float(((a - b).abs().amax(dim=2) / b.abs().amax(dim=2).clamp_min(1e-30)).max())


worst = 0.0
done = 0
while done < cases:
    W = (rng.choice([2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 40]), rng.choice([2, 4, 6, 8, 12, 16, 24, 32, 36]))
    G = W[0] * W[1]
    N = 4 * rng.randint(1, 1024)
    C = 4 * rng.randint(1, 9)
    B, H = rng.randint(1, 3), rng.randint(1, 5)
    if G % 4 or N // 4 + 3 * G // 4 > 2048:
        continue
    pad = rng.random() < 0.3
    torch.manual_seed(rng.randint(0, 1 << 30))
    spread = rng.choice([0.2, 1.0, 3.0])
    keys = torch.tanh(torch.randn(B, H * 2, N, device="cuda") * spread)
    if rng.random() < 0.3:
        keys[:, :, N // 2:] = keys[:, :, :N - N // 2]
    z = torch.randn(B, H * C, *W, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda") * (10.0 ** rng.uniform(-3, 3))
    p = (torch.rand(B, N, device="cuda") > 0.3).float() if pad else None
    Wa = _lib.int_array(list(W))
    nws = max(lib.ct_slice_bwd_workspace_bytes(B, H, C, N, 2, Wa), 2 * keys.numel() * 4, 16)
    ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
    tk = torch.zeros(_lib.TICKETS_BYTES // 4, device="cuda", dtype=torch.int32) if rng.random() < 0.5 else None
    use_ws = rng.random() < 0.7
    res = {}
    for name, fl in (("scatter", _lib.DEBUG_NO_SORTED | _lib.DEBUG_FORCE_HOT), ("sorted", _lib.DEBUG_FORCE_SORTED | _lib.DEBUG_FORCE_HOT)):
        g_z, g_k = torch.full_like(z, float("nan")), torch.full_like(keys, float("nan"))
        lib.ct_debug_set_flags(fl)
        rc = lib.ct_slice_bwd_ps(_ptr(keys), _ptr(z), _ptr(p), _lib.PAD_F32 if pad else 0, _ptr(cot), _ptr(g_z), _ptr(g_k),
                                 _ptr(ws) if use_ws else None, nws if use_ws else 0, _ptr(tk), None, B, H, C, N, 2, Wa, _stream())
        tag = lib.ct_debug_last_launch().decode()
        zs = ops.splat_keys(keys, cot, p, list(W), H, 2, "sum")
        tag2 = lib.ct_debug_last_launch().decode()
        lib.ct_debug_set_flags(0)
        assert rc == 0, (rc, name)
        res[name] = (g_z, g_k, zs, tag, tag2)
    torch.cuda.synchronize()
    if not res["sorted"][3].startswith("slice_bwd_sorted"):        # (the carve-up does not fit a CU's LDS: not a sorted layout)
        continue
    assert res["sorted"][4] == "scatter_add_sorted", res["sorted"][4]
    if tk is not None:
        assert int(tk.abs().sum()) == 0
    # float64 scatter-add of the same fp32 products' factors: which form is nearer the truth where they disagree
    lc, idx = ops.positions(keys, list(W), H, 2)                                  # (B, H, 4, N)
    src = (cot * p[:, None, :]) if pad else cot
    pre = (src.double().reshape(B, H, C, 1, N) * lc.double().reshape(B, H, 1, 4, N)).reshape(B, H, C, 4 * N)
    ref = torch.zeros(B, H, C, G, dtype=torch.float64, device="cuda").scatter_add_(3, idx.reshape(B, H, 1, 4 * N).expand(B, H, C, 4 * N), pre)
    ref = ref.reshape(B, H * C, *W)
    e64 = (per_channel_err(res["sorted"][0], ref), per_channel_err(res["scatter"][0], ref))
    e = (min(e64[0], per_channel_err(res["sorted"][0], res["scatter"][0])), float((res["sorted"][1] - res["scatter"][1]).abs().max() / res["scatter"][1].abs().max().clamp_min(1e-30)),
         min(per_channel_err(res["sorted"][2], ref), per_channel_err(res["sorted"][2], res["scatter"][2])))
    worst64 = max(globals().get("worst64", 0.0), e64[0]); globals()["worst64"] = worst64
    worst = max(worst, *e)
    if max(e) > 1e-4 or not all(torch.isfinite(t).all() for t in res["sorted"][:3]):
        print("FAIL", (B, H, C, N, W, pad, use_ws, tk is not None), res["sorted"][3], e, "vs float64: sorted %.2e scatter %.2e" % e64, flush=True)
        sys.exit(1)
    done += 1
print("%d cases, worst error %.2e (sorted g_grid vs float64: %.2e)" % (done, worst, worst64))

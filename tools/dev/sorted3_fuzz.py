"""Random layouts through the sorted-SEGMENT kernels of round 6 (csrc/ct_raster_sorted3d.h): Slice backward on 3D grids of <= 1024
cells and on small 2D grids (the segment form forced), Splat(sum) forward on small 3D grids — each against the scatter form on the
same inputs and against a float64 scatter-add of the same products.  python tools/dev/sorted3_fuzz.py [cases] [seed]"""
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


worst = worst64 = 0.0
done = skipped = 0
tags = {}
while done < cases:
    dim = 3 if rng.random() < 0.7 else 2
    if dim == 3:
        W = tuple(rng.choice([2, 3, 4, 5, 6, 8, 10, 12, 16]) for _ in range(3))
    else:
        W = (rng.choice([2, 4, 6, 8, 12, 16, 20, 24]), rng.choice([2, 4, 8, 12, 16, 24, 32]))
    G = 1
    for w in W:
        G *= w
    if G % 4 or G > 1024:
        continue
    N = 4 * rng.randint(1, 2048 if rng.random() < 0.8 else 5000)
    C = 4 * rng.randint(1, 10)
    B, H = rng.randint(1, 3), rng.randint(1, 5)
    pad = rng.random() < 0.3
    torch.manual_seed(rng.randint(0, 1 << 30))
    spread = rng.choice([0.2, 1.0, 3.0])
    keys = torch.tanh(torch.randn(B, H * dim, N, device="cuda") * spread)
    if rng.random() < 0.3:
        keys[:, :, N // 2:] = keys[:, :, :N - N // 2]
    if rng.random() < 0.2:          # clamp edges and points on cell boundaries
        keys[0, :, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.99999994], device="cuda")
    z = torch.randn(B, H * C, *W, device="cuda")
    cot = torch.randn(B, H * C, N, device="cuda") * (10.0 ** rng.uniform(-3, 3))
    p = (torch.rand(B, N, device="cuda") > 0.3).float() if pad else None
    Wa = _lib.int_array(list(W))
    tk = torch.zeros(_lib.TICKETS_BYTES // 4, device="cuda", dtype=torch.int32) if rng.random() < 0.5 else None
    res = {}
    forced = _lib.DEBUG_FORCE_SORTED | _lib.DEBUG_FORCE_HOT | (_lib.DEBUG_FORCE_SORTED_SEG if dim == 2 else 0)
    for name, fl in (("scatter", _lib.DEBUG_NO_SORTED | _lib.DEBUG_FORCE_HOT), ("sorted", forced)):
        lib.ct_debug_set_flags(fl)
        nws = max(lib.ct_slice_bwd_workspace_bytes(B, H, C, N, dim, Wa), 16)
        ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
        g_z, g_k = torch.full_like(z, float("nan")), torch.full_like(keys, float("nan"))
        rc = lib.ct_slice_bwd_ps(_ptr(keys), _ptr(z), _ptr(p), _lib.PAD_F32 if pad else 0, _ptr(cot), _ptr(g_z), _ptr(g_k),
                                 _ptr(ws), nws, _ptr(tk), None, B, H, C, N, dim, Wa, _stream())
        tag = lib.ct_debug_last_launch().decode()
        zs = ops.splat_keys(keys, cot, p, list(W), H, dim, "sum")
        tag2 = lib.ct_debug_last_launch().decode()
        lib.ct_debug_set_flags(0)
        assert rc == 0, (rc, name, (B, H, C, N, W))
        res[name] = (g_z, g_k, zs, tag, tag2)
    torch.cuda.synchronize()
    stag = res["sorted"][3]
    if not (stag.startswith("slice_bwd_sorted3") or stag.startswith("slice_bwd_sorted2s")):
        skipped += 1          # the plan declined the layout (LDS): not a sorted-segment launch
        continue
    tags[stag] = tags.get(stag, 0) + 1
    tags[res["sorted"][4]] = tags.get(res["sorted"][4], 0) + 1
    if tk is not None:
        assert int(tk.abs().sum()) == 0, "tickets not handed back as zeros"
    V = 1 << dim
    lc, idx = ops.positions(keys, list(W), H, dim)
    src = (cot * p[:, None, :]) if pad else cot
    pre = (src.double().reshape(B, H, C, 1, N) * lc.double().reshape(B, H, 1, V, N)).reshape(B, H, C, V * N)
    ref = torch.zeros(B, H, C, G, dtype=torch.float64, device="cuda").scatter_add_(3, idx.reshape(B, H, 1, V * N).expand(B, H, C, V * N), pre)
    ref = ref.reshape(B, H * C, *W)
    e64 = (per_channel_err(res["sorted"][0], ref), per_channel_err(res["scatter"][0], ref))
    e = (min(e64[0], per_channel_err(res["sorted"][0], res["scatter"][0])),
         float((res["sorted"][1] - res["scatter"][1]).abs().max() / res["scatter"][1].abs().max().clamp_min(1e-30)),
         min(per_channel_err(res["sorted"][2], ref), per_channel_err(res["sorted"][2], res["scatter"][2])))
    worst64 = max(worst64, e64[0])
    worst = max(worst, *e)
    if max(e) > 1e-4 or not all(torch.isfinite(t).all() for t in res["sorted"][:3]):
        print("FAIL", (B, H, C, N, W, pad, tk is not None), stag, res["sorted"][4], e, "vs float64: sorted %.2e scatter %.2e" % e64, flush=True)
        sys.exit(1)
    done += 1
print("%d cases (%d layouts declined by the plan), worst error %.2e (sorted g_grid vs float64: %.2e)" % (done, skipped, worst, worst64))
print("launch tags:", dict(sorted(tags.items())))

"""torch.autograd.Function wrappers over the C ABI (include/cloudct.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream;
every computation happens inside libcloudct.so.  Tensors that are not on a HIP
device raise — there is no CPU path in the product.
"""
import ctypes
import math
import os

import torch

from . import _lib


def sizes_of(tensor_size, dim):
    """int -> [W]*dim; tuple/list of len dim kept (layers/cloud_transform.py:41-46)."""
    if isinstance(tensor_size, int):
        return [tensor_size] * dim
    sizes = [int(w) for w in tensor_size]
    assert len(sizes) == dim
    return sizes


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "cloud_transformers_amd ops need tensors on a HIP device (MI355X); "
                "got a %s tensor and there is no CPU fallback" % t.device.type)


def _stream(device=None):
    """The raw hipStream_t the launch goes to: torch's current stream of `device` (default: the current device).  Through the two
    C entry points directly: torch.cuda.current_stream().cuda_stream builds a Stream object and parses a device argument on
    every call, 9 us of host time against 0.4 — at 250-400 launches per eager model step that was 2-3 ms of a host-bound step
    (tools/dev/host_profile.py)."""
    idx = device.index if device is not None and device.index is not None else _cuda_get_device()
    return _cuda_raw_stream(idx)


# (private entry points of torch._C, present since torch 1.x; a build without them takes the public, slower route)
_cuda_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_cuda_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context that makes `device` current for the launch: a no-op object when it already is (the usual case;
    half the host cost of torch.cuda.device(), 0.6 vs 1.4 us per op)."""
    if device.index is None or device.index == _cuda_get_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    return t.contiguous()


def _pad_args(pad, B, N):
    """pts_padding (B,N) float or int32 (datasets/s3dis_closer.py:330) -> (tensor, dtype code)."""
    if pad is None:
        return None, _lib.PAD_NONE
    assert pad.shape == (B, N), "pts_padding must be [batch, num_points]"
    if pad.dtype == torch.float32:
        return pad.contiguous(), _lib.PAD_F32
    if pad.dtype == torch.int32:
        return pad.contiguous(), _lib.PAD_I32
    return pad.to(torch.float32).contiguous(), _lib.PAD_F32


# ---------------------------------------------------------------------------
# Pointwise-convolution GEMMs (ct_pw_gemm: fp32 in / out, split-f16 on the matrix pipes)
# ---------------------------------------------------------------------------
PW_FWD, PW_DGRAD, PW_WGRAD, PW_DGRAD_T = 0, 1, 2, 3
# "split16": this library's kernels; "lib": rocBLAS fp32 through torch.bmm (the round-2 path, kept as the A/B switch and
# for shapes the kernel does not take: a dimension that is not a multiple of 4)
PW_GEMM = os.environ.get("CLOUDCT_PW_GEMM", "split16")


PW_BWD_STREAMS = os.environ.get("CLOUDCT_PW_BWD_STREAMS", "1") != "0"
_pw_side = {}
# streams that are themselves forks inside a capture (the heads' side streams of layers.multihead_ct._run_heads): work on them
# does not fork again — a second level of forks shared by two concurrent chains crashed hipStreamEndCapture
forked_streams = set()


def _pw_side_stream(device):
    s = _pw_side.get(device.index)
    if s is None:
        s = _pw_side[device.index] = torch.cuda.Stream(device=device)
    return s


def pw_eligible(Co, Ci, N, mode=None):
    """Whether ct_pw_gemm takes the product; with `mode`, whether it also beats the library GEMM there: an output-row
    extent under one 128-row tile leaves half the MFMA rows idle (tools/pw_gemm_bench.py: 0.7-0.95x at 64 rows)."""
    if not (PW_GEMM == "split16" and Co % 4 == 0 and Ci % 4 == 0 and N % 4 == 0 and N >= 4):
        return False
    if mode == PW_FWD:
        return Co >= 128
    if mode == PW_DGRAD:
        return Ci >= 128
    if mode == PW_WGRAD:
        return Co >= 128 and Ci >= 128
    return True


_amax_len = None


def amax(t):
    """Partial maxima of |t| for a contiguous float32 tensor: f32[ct_amax_len()], one per block of the kernel (max |t| is
    their maximum; ct_pw_gemm folds them when it starts) — the per-tensor scale of ct_pw_gemm."""
    global _amax_len
    lib = _lib.load()
    if _amax_len is None:
        _amax_len = lib.ct_amax_len()
    out = torch.empty(_amax_len, device=t.device, dtype=torch.float32)
    with _on(t.device):
        _lib.check(lib.ct_amax_f32(_ptr(t), t.numel(), _ptr(out), _stream()), "ct_amax_f32")
    return out


PW_AMAX_MAX = 32768     # partial maxima ct_pw_gemm folds per operand


def _amax_slots(C, device):
    """Maxima buffer for a producer kernel (ct_bn_relu_*_amax: one slot per channel; ct_adain_*_amax: one per (cloud, channel)),
    or None when the consumer could not use it (library GEMMs selected, or more slots than ct_pw_gemm folds)."""
    if PW_GEMM != "split16" or C > PW_AMAX_MAX:
        return None
    return torch.empty(C, device=device, dtype=torch.float32)


def tag_amax(t, slots):
    """Remember on tensor `t` [B, C, N] the maxima its producer left in `slots` (valid until `t` is modified in place).  The
    producers write one maximum per channel (or per (cloud, channel)): kept as a 2-D [n / C, C] view — the shape is what tells
    ct_pw_gemm_rs that these are PER-ROW maxima (a 1-D tensor holds partials of one maximum, ops.amax)."""
    if slots is not None:
        C = t.shape[1] if t.dim() == 3 else 0
        if C and slots.dim() == 1 and slots.numel() % C == 0:
            slots = slots.view(-1, C)
        t._ct_amax = (slots, t._version)
    return t


def amax_rows(t):
    """Per-channel maxima of |t| for a contiguous float32 [B, C, N] tensor, as a [1, C] tensor (ct_amax_rows_f32): what the
    weight gradient wants of an operand no producer left maxima for."""
    B, C, N = t.shape
    if C > PW_AMAX_MAX or N % 4 != 0:
        return amax(t)
    out = torch.empty(1, C, device=t.device, dtype=torch.float32)
    with _on(t.device):
        _lib.check(_lib.load().ct_amax_rows_f32(_ptr(t), B, C, N, _ptr(out), _stream()), "ct_amax_rows_f32")
    return out


def _rows_of(am, want):
    """`want` when `am` holds per-row maxima of an operand with `want` rows (2-D [.., want]), else 0 (partials of one maximum)."""
    return want if (am is not None and am.dim() == 2 and am.shape[1] == want) else 0


def amax_of(t, rows=False):
    """The operand maxima of `t` for ct_pw_gemm: what its producer left behind if `t` is unchanged since (an in-place op
    bumps `_version`), else a pass over it — ct_amax_rows_f32 (per channel) where the consumer can use per-row scales
    (`rows`: the weight gradient), ct_amax_f32 otherwise."""
    tag = getattr(t, "_ct_amax", None)
    if tag is not None and tag[1] == t._version and tag[0].device == t.device:
        return tag[0]
    return amax_rows(t) if (rows and t.dim() == 3) else amax(t)


def pw_gemm(mode, a, b, amax_a, amax_b, B, Co, Ci, N, addend=None):
    """ct_pw_gemm on contiguous float32 tensors: PW_FWD (a = W [Co,Ci], b = x [B,Ci,N]) -> y [B,Co,N]; PW_DGRAD (a = W,
    b = g_y [B,Co,N]) -> g_x [B,Ci,N]; PW_WGRAD (a = g_y, b = x) -> g_W [Co,Ci].  amax_* from amax().  addend (contiguous,
    shaped as the output; not for PW_WGRAD): added in the kernel's epilogue (ct_pw_gemm_rs_add)."""
    lib = _lib.load()
    dev = b.device
    if mode == PW_FWD:
        out = torch.empty(B, Co, N, device=dev, dtype=torch.float32)
    elif mode in (PW_DGRAD, PW_DGRAD_T):
        out = torch.empty(B, Ci, N, device=dev, dtype=torch.float32)
    else:
        out = torch.empty(Co, Ci, device=dev, dtype=torch.float32)
    nbytes = lib.ct_pw_gemm_workspace_bytes(mode, B, Co, Ci, N)
    ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32) if nbytes else None
    # per-row scales wherever the maxima are per row of the operand's k-contiguous arrangement (2-D: tag_amax, prep_weight)
    rows_a = _rows_of(amax_a, Co if mode in (PW_FWD, PW_WGRAD) else Ci) if mode != PW_DGRAD else 0
    rows_b = _rows_of(amax_b, Ci) if mode == PW_WGRAD else 0
    if addend is not None:
        assert mode != PW_WGRAD and addend.shape == out.shape and addend.is_contiguous() and addend.dtype == torch.float32
    with _on(dev):
        _lib.check(lib.ct_pw_gemm_rs_add(mode, _ptr(a), _ptr(b), _ptr(out), _ptr(addend), _ptr(amax_a),
                                         0 if amax_a is None else amax_a.numel(), rows_a, _ptr(amax_b),
                                         0 if amax_b is None else amax_b.numel(), rows_b, _ptr(ws), nbytes, B, Co, Ci, N, _stream()),
                   "ct_pw_gemm_rs_add")
    return out


def prep_weight(W, transpose):
    """((row maxima of |W|, column maxima), W^T or None) in ONE launch (ct_pw_prep_weight_rs): what the three products of a layer
    need of its weight — per-row scales for W in the forward ([tiles, Co]) and for W^T in the data gradient ([tiles, Ci]); falls
    back to (amax(), None) and no transpose for weights with more maxima than ct_pw_gemm_rs folds."""
    lib = _lib.load()
    Co, Ci = W.shape
    tc, tr = (Ci + 31) // 32, (Co + 31) // 32
    if tc * Co > PW_AMAX_MAX or tr * Ci > PW_AMAX_MAX:
        return (amax(W), None), None
    rowmax = torch.empty(tc, Co, device=W.device, dtype=torch.float32)
    colmax = torch.empty(tr, Ci, device=W.device, dtype=torch.float32)
    Wt = torch.empty(Ci, Co, device=W.device, dtype=torch.float32) if transpose else None
    with _on(W.device):
        _lib.check(lib.ct_pw_prep_weight_rs(_ptr(W), _ptr(Wt), _ptr(rowmax), _ptr(colmax), Co, Ci, _stream()), "ct_pw_prep_weight_rs")
    return (rowmax, colmax), Wt


def pw_forward(W, x, need_dgrad=False):
    """y[b] = W x[b] for W [Co,Ci], x [B,Ci,N] (both contiguous float32 on the device); returns (y, amax_W, amax_x, W^T) — the
    maxima (amax_W: the pair (row maxima, column maxima) of prep_weight) and, with need_dgrad, the transposed weight are reused
    by the gradients — or (y, None, None, None) from the library GEMM."""
    Co, Ci = W.shape
    B, _, N = x.shape
    if not pw_eligible(Co, Ci, N, PW_FWD):
        return torch.bmm(W.unsqueeze(0).expand(B, -1, -1), x), None, None, None
    am_w, Wt = prep_weight(W, need_dgrad and pw_eligible(Co, Ci, N, PW_DGRAD))
    am_x = amax_of(x)
    return pw_gemm(PW_FWD, W, x, am_w[0], am_x, B, Co, Ci, N), am_w, am_x, Wt


def pw_backward(W, x, g_y, am_w, am_x, need_x=True, need_w=True, am_g=None, Wt=None, add_gx=None):
    """(g_x, g_W) of pw_forward for the cotangent g_y [B,Co,N] (contiguous); am_g: g_y's maxima where the caller has them; Wt:
    the transposed weight pw_forward made (else the data gradient writes its own through its workspace); add_gx: another
    cotangent of x (a skip connection's), added to g_x inside the data gradient's epilogue where the kernel runs it."""
    Co, Ci = W.shape
    B, _, N = x.shape
    mine_x = need_x and pw_eligible(Co, Ci, N, PW_DGRAD)
    mine_w = need_w and pw_eligible(Co, Ci, N, PW_WGRAD)
    if am_g is None and (mine_x or mine_w):
        am_g = amax_of(g_y, rows=mine_w)
    if torch.is_tensor(am_w):            # (a caller that kept one maxima tensor of W: partials of its maximum)
        am_w = (am_w, None)
    g_x = g_w = None

    if add_gx is not None:
        add_gx = _f32c(add_gx)

    def dgrad():
        if mine_x and Wt is not None and am_w is not None:
            # W^T's rows are W's columns: their maxima give the data gradient its per-row scales
            return pw_gemm(PW_DGRAD_T, Wt, g_y, am_w[1] if am_w[1] is not None else am_w[0], am_g, B, Co, Ci, N, addend=add_gx)
        if mine_x:
            return pw_gemm(PW_DGRAD, W, g_y, am_w[0] if am_w is not None else amax(W), am_g, B, Co, Ci, N, addend=add_gx)
        if not need_x:
            return add_gx
        g = torch.bmm(W.t().unsqueeze(0).expand(B, -1, -1), g_y)
        return g if add_gx is None else g + add_gx

    def wgrad():
        if mine_w:
            # both operands row-scaled where their maxima are per channel: a nearly-dead channel of g_y (or of x) keeps its
            # 22 bits in its own row of g_W (include/cloudct.h, ct_pw_gemm_rs)
            return pw_gemm(PW_WGRAD, g_y, x, am_g, am_x if am_x is not None else amax_of(x, rows=True), B, Co, Ci, N)
        return torch.bmm(g_y, x.transpose(1, 2)).sum(0) if need_w else None

    # the two gradients are independent and neither is a whole number of rounds of the chip's workgroup slots (1024 + 672
    # workgroups on 512 slots at 848 x 512): inside a HIP-graph capture they go to two streams and fill each other's tails
    if (PW_BWD_STREAMS and mine_x and mine_w and g_y.is_cuda and torch.cuda.is_current_stream_capturing()
            and _stream(g_y.device) not in forked_streams):
        cur = torch.cuda.current_stream(g_y.device)
        side = _pw_side_stream(g_y.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g_x = dgrad()
        g_w = wgrad()
        cur.wait_stream(side)
        g_x.record_stream(cur)
    else:
        g_x, g_w = dgrad(), wgrad()
    return g_x, g_w


# ---------------------------------------------------------------------------
# DifferentiablePositions
# ---------------------------------------------------------------------------
class PositionsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, keys, W, H):
        _dev(keys)
        keys = _f32c(keys)
        dim = len(W)
        B, HD, N = keys.shape
        assert HD == H * dim
        V = 1 << dim
        lc = torch.empty(B, H, V, N, device=keys.device, dtype=torch.float32)
        idx = torch.empty(B, H, V, N, device=keys.device, dtype=torch.int64)
        lib = _lib.load()
        with _on(keys.device):
            _lib.check(lib.ct_positions_fwd(_ptr(keys), _ptr(lc), _ptr(idx), B, H, N, dim,
                                            _lib.int_array(W), _stream()), "ct_positions_fwd")
        ctx.save_for_backward(keys)
        ctx.W, ctx.H = W, H
        ctx.mark_non_differentiable(idx)
        return lc, idx

    @staticmethod
    def backward(ctx, g_lc, _g_idx):
        (keys,) = ctx.saved_tensors
        W, H = ctx.W, ctx.H
        B, _, N = keys.shape
        g_lc = _f32c(g_lc)
        g_keys = torch.empty_like(keys)
        lib = _lib.load()
        with _on(keys.device):
            _lib.check(lib.ct_positions_bwd(_ptr(keys), _ptr(g_lc), _ptr(g_keys), B, H, N, len(W),
                                            _lib.int_array(W), _stream()), "ct_positions_bwd")
        return g_keys, None, None


# ---------------------------------------------------------------------------
# fused (keys-based) Splat / Slice — the hot path
# ---------------------------------------------------------------------------
RASTER_TICKETS = os.environ.get("CLOUDCT_TICKETS", "1") != "0"      # "0": the two-launch form (A/B, debugging)
_tickets = {}


def raster_tickets(device, force=False):
    """The arrival tickets of the backward raster passes (include/cloudct.h: ct_slice_bwd_tk / ct_splat_bwd_tk) for the
    CURRENT stream of `device`: a zeroed CT_TICKETS_BYTES buffer kept for the life of the process.  The kernels leave it zero,
    so it is initialised once; launches that share a buffer must be ordered, hence one buffer per stream (the heads of a
    union block run side by side on their own streams, and autograd replays every backward on its forward's stream)."""
    if not RASTER_TICKETS and not force:
        return None
    key = (device.index, _stream(device))
    t = _tickets.get(key)
    if t is None:
        t = _tickets[key] = torch.zeros(_lib.TICKETS_BYTES // 4, device=device, dtype=torch.int32)
    return t


class SplatKeysFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, keys, feat, pad, W, H, reduce):
        _dev(keys, feat, pad)
        keys, feat = _f32c(keys), _f32c(feat)
        dim = len(W)
        B, HC, N = feat.shape
        assert HC % H == 0 and keys.shape == (B, H * dim, N)
        C = HC // H
        padt, pad_code = _pad_args(pad, B, N)
        grid = torch.empty(B, HC, *W, device=feat.device, dtype=torch.float32)
        lib = _lib.load()
        with _on(feat.device):
            _lib.check(lib.ct_splat_fwd(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(grid),
                                        B, H, C, N, dim, _lib.int_array(W), _lib.REDUCE[reduce], _stream()),
                       "ct_splat_fwd")
        ctx.save_for_backward(keys, feat, padt, grid if reduce == "max" else None)
        ctx.meta = (W, H, C, reduce, pad_code)
        return grid

    @staticmethod
    def backward(ctx, g_grid):
        keys, feat, padt, grid = ctx.saved_tensors
        W, H, C, reduce, pad_code = ctx.meta
        dim = len(W)
        B, HC, N = feat.shape
        g_grid = _f32c(g_grid)
        g_feat = torch.empty_like(feat)
        g_keys = torch.empty_like(keys)
        lib = _lib.load()
        Wa = _lib.int_array(W)
        ws_bytes = lib.ct_splat_bwd_workspace_bytes(B, H, C, N, dim, Wa, _lib.REDUCE[reduce])
        ws = torch.empty(ws_bytes, device=feat.device, dtype=torch.uint8) if ws_bytes else None
        with _on(feat.device):
            _lib.check(lib.ct_splat_bwd_tk(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(grid), _ptr(g_grid),
                                           _ptr(g_feat), None, _ptr(g_keys), _ptr(ws), ws_bytes, _ptr(raster_tickets(feat.device)),
                                           B, H, C, N, dim, Wa, _lib.REDUCE[reduce], _stream()), "ct_splat_bwd_tk")
        return g_keys, g_feat, None, None, None, None


class SliceKeysFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, keys, grid, pad, W, H):
        _dev(keys, grid, pad)
        keys, grid = _f32c(keys), _f32c(grid)
        dim = len(W)
        B, HC = grid.shape[:2]
        N = keys.shape[-1]
        assert HC % H == 0 and keys.shape == (B, H * dim, N) and list(grid.shape[2:]) == list(W)
        C = HC // H
        padt, pad_code = _pad_args(pad, B, N)
        out = torch.empty(B, HC, N, device=grid.device, dtype=torch.float32)
        lib = _lib.load()
        with _on(grid.device):
            _lib.check(lib.ct_slice_fwd(_ptr(keys), _ptr(grid), _ptr(padt), pad_code, _ptr(out),
                                        B, H, C, N, dim, _lib.int_array(W), _stream()), "ct_slice_fwd")
        ctx.save_for_backward(keys, grid, padt)
        ctx.meta = (W, H, C, pad_code)
        return out

    @staticmethod
    def backward(ctx, g_out):
        keys, grid, padt = ctx.saved_tensors
        W, H, C, pad_code = ctx.meta
        dim = len(W)
        B, _, N = keys.shape
        g_out = _f32c(g_out)
        g_grid = torch.empty_like(grid)
        g_keys = torch.empty_like(keys)
        lib = _lib.load()
        Wa = _lib.int_array(W)
        ws_bytes = lib.ct_slice_bwd_workspace_bytes(B, H, C, N, dim, Wa)
        ws = torch.empty(ws_bytes, device=grid.device, dtype=torch.uint8) if ws_bytes else None
        with _on(grid.device):
            _lib.check(lib.ct_slice_bwd_tk(_ptr(keys), _ptr(grid), _ptr(padt), pad_code, _ptr(g_out),
                                           _ptr(g_grid), _ptr(g_keys), _ptr(ws), ws_bytes, _ptr(raster_tickets(grid.device)),
                                           B, H, C, N, dim, Wa, _stream()),
                       "ct_slice_bwd_tk")
        return g_keys, g_grid, None, None, None


# ---------------------------------------------------------------------------
# explicit (local_coordinate, flattened_index) Splat / Slice — API-compatible path
# ---------------------------------------------------------------------------
def _check_idx(idx):
    if idx.dtype != torch.int64:
        raise TypeError("flattened_index must be int64 (layers/cloud_transform.py:81)")
    return idx.contiguous()


class SplatLcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lc, idx, feat, pad, W, H, reduce):
        _dev(lc, idx, feat, pad)
        lc, feat, idx = _f32c(lc), _f32c(feat), _check_idx(idx)
        dim = len(W)
        B, HC, N = feat.shape
        V = 1 << dim
        assert HC % H == 0 and lc.shape == (B, H, V, N) and idx.shape == (B, H, V, N)
        C = HC // H
        padt, pad_code = _pad_args(pad, B, N)
        grid = torch.empty(B, HC, *W, device=feat.device, dtype=torch.float32)
        lib = _lib.load()
        with _on(feat.device):
            _lib.check(lib.ct_splat_lc_fwd(_ptr(lc), _ptr(idx), _ptr(feat), _ptr(padt), pad_code, _ptr(grid),
                                           B, H, C, N, dim, _lib.int_array(W), _lib.REDUCE[reduce], _stream()),
                       "ct_splat_lc_fwd")
        ctx.save_for_backward(lc, idx, feat, padt, grid if reduce == "max" else None)
        ctx.meta = (W, H, C, reduce, pad_code)
        return grid

    @staticmethod
    def backward(ctx, g_grid):
        lc, idx, feat, padt, grid = ctx.saved_tensors
        W, H, C, reduce, pad_code = ctx.meta
        dim = len(W)
        B, HC, N = feat.shape
        g_grid = _f32c(g_grid)
        g_feat = torch.empty_like(feat)
        g_lc = torch.empty_like(lc)
        lib = _lib.load()
        Wa = _lib.int_array(W)
        ws_bytes = lib.ct_splat_bwd_workspace_bytes(B, H, C, N, dim, Wa, _lib.REDUCE[reduce])
        ws = torch.empty(ws_bytes, device=feat.device, dtype=torch.uint8) if ws_bytes else None
        with _on(feat.device):
            _lib.check(lib.ct_splat_lc_bwd(_ptr(lc), _ptr(idx), _ptr(feat), _ptr(padt), pad_code, _ptr(grid),
                                           _ptr(g_grid), _ptr(g_feat), _ptr(g_lc), _ptr(ws), ws_bytes,
                                           B, H, C, N, dim, Wa, _lib.REDUCE[reduce], _stream()), "ct_splat_lc_bwd")
        return g_lc, None, g_feat, None, None, None, None


class SliceLcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lc, idx, grid, pad, W, H):
        _dev(lc, idx, grid, pad)
        lc, grid, idx = _f32c(lc), _f32c(grid), _check_idx(idx)
        dim = len(W)
        B, HC = grid.shape[:2]
        V = 1 << dim
        N = lc.shape[-1]
        assert HC % H == 0 and lc.shape == (B, H, V, N) and idx.shape == (B, H, V, N)
        assert math.prod(grid.shape[2:]) == math.prod(W)
        C = HC // H
        padt, pad_code = _pad_args(pad, B, N)
        out = torch.empty(B, HC, N, device=grid.device, dtype=torch.float32)
        lib = _lib.load()
        with _on(grid.device):
            _lib.check(lib.ct_slice_lc_fwd(_ptr(lc), _ptr(idx), _ptr(grid), _ptr(padt), pad_code, _ptr(out),
                                           B, H, C, N, dim, _lib.int_array(W), _stream()), "ct_slice_lc_fwd")
        ctx.save_for_backward(lc, idx, grid, padt)
        ctx.meta = (W, H, C, pad_code)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lc, idx, grid, padt = ctx.saved_tensors
        W, H, C, pad_code = ctx.meta
        dim = len(W)
        B, _, _, N = lc.shape
        g_out = _f32c(g_out)
        g_grid = torch.empty_like(grid)
        g_lc = torch.empty_like(lc)
        lib = _lib.load()
        with _on(grid.device):
            _lib.check(lib.ct_slice_lc_bwd(_ptr(lc), _ptr(idx), _ptr(grid), _ptr(padt), pad_code, _ptr(g_out),
                                           _ptr(g_grid), _ptr(g_lc), B, H, C, N, dim, _lib.int_array(W), _stream()),
                       "ct_slice_lc_bwd")
        return g_lc, None, g_grid, None, None, None


# ---------------------------------------------------------------------------
# lattice: per-head rigid transform of (xyz + residual) and tanh, fused
# ---------------------------------------------------------------------------
class LatticeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, residual, R, shift, scales, kscale, dim, with_stats=False):
        _dev(xyz, residual, R, shift, scales, kscale)
        xyz, residual, R, shift = _f32c(xyz), _f32c(residual), _f32c(R), _f32c(shift)
        scales = _f32c(scales) if scales is not None else None
        ks = _f32c(kscale).reshape(1) if kscale is not None else None
        B, _, N = xyz.shape
        H = R.shape[0]
        assert xyz.shape[1] == 3 and residual.shape == (B, H * 3, N) and R.shape == (H, 3, 3) and shift.shape == (H, 3)
        keys = torch.empty(B, H * dim, N, device=xyz.device, dtype=torch.float32)
        lattice = torch.empty_like(keys)
        lib = _lib.load()
        stats = ws = None
        ws_bytes = 0
        if with_stats:
            stats = torch.empty(2, device=xyz.device, dtype=torch.float32)
            ws_bytes = lib.ct_lattice_fwd_workspace_bytes(B, H, N)
            ws = torch.empty(ws_bytes, device=xyz.device, dtype=torch.uint8)
        with _on(xyz.device):
            _lib.check(lib.ct_lattice_fwd(_ptr(xyz), _ptr(residual), _ptr(R), _ptr(shift), _ptr(scales), _ptr(ks),
                                          _ptr(keys), _ptr(lattice), _ptr(stats), _ptr(ws), ws_bytes, B, H, N, dim, _stream()),
                       "ct_lattice_fwd")
        ctx.save_for_backward(xyz, residual, R, shift, scales, ks, lattice)
        ctx.meta = (B, H, N, dim, kscale.shape if kscale is not None else None)
        # a block uses one of (keys, tanh(keys)) downstream: the other's cotangent stays None instead of a zero tensor that a
        # fill kernel writes and the backward kernel reads (ct_lattice_bwd takes either pointer as NULL)
        ctx.set_materialize_grads(False)
        if with_stats:
            ctx.mark_non_differentiable(stats)
            return keys, lattice, stats
        return keys, lattice

    @staticmethod
    def backward(ctx, g_keys, g_lattice, _g_stats=None):
        xyz, residual, R, shift, scales, ks, lattice = ctx.saved_tensors
        B, H, N, dim, ks_shape = ctx.meta
        if g_keys is None and g_lattice is None:
            return (None,) * 8
        g_keys = _f32c(g_keys) if g_keys is not None else None
        g_lattice = _f32c(g_lattice) if g_lattice is not None else None
        g_xyz, g_res = torch.empty_like(xyz), torch.empty_like(residual)
        g_R, g_shift = torch.empty_like(R), torch.empty_like(shift)
        g_scales = torch.empty_like(scales) if scales is not None else None
        g_ks = torch.empty_like(ks) if ks is not None else None
        lib = _lib.load()
        with _on(xyz.device):
            ws_bytes = lib.ct_lattice_bwd_workspace_bytes(B, H, N)
            ws = torch.empty(ws_bytes, device=xyz.device, dtype=torch.uint8)
            _lib.check(lib.ct_lattice_bwd(_ptr(xyz), _ptr(residual), _ptr(R), _ptr(shift), _ptr(scales), _ptr(ks),
                                          _ptr(lattice), _ptr(g_lattice), _ptr(g_keys), _ptr(g_xyz), _ptr(g_res),
                                          _ptr(g_R), _ptr(g_shift), _ptr(g_scales), _ptr(g_ks), _ptr(ws), ws_bytes,
                                          B, H, N, dim, _stream()),
                       "ct_lattice_bwd")
        return g_xyz, g_res, g_R, g_shift, g_scales, (g_ks.reshape(ks_shape) if g_ks is not None else None), None, None


def lattice(xyz, residual, R, shift, scales, kscale, dim, with_stats=False):
    """(keys, tanh(keys)) of an MHCT block; R = so3_exponential_map(log_R) [H,3,3].  with_stats: also a
    non-differentiable f32[2] = (mean, unbiased variance) of the keys, reduced inside the same launch."""
    return LatticeFn.apply(xyz, residual, R, shift, scales, kscale, dim, with_stats)


class LatticeSo3Fn(torch.autograd.Function):
    """LatticeFn with the so3 exponential map of the rotation parameters inside the launches (ct_lattice_so3_fwd / _bwd): the
    forward is ONE launch (map + transform + tanh + key statistics, finished by the launch's last workgroup), the backward two
    (the tail launch also turns g_R into g_log_R) — three launches per head and step where LatticeFn + So3ExpFn are six."""

    @staticmethod
    def forward(ctx, xyz, residual, log_R, shift, scales, kscale, dim, eps, with_stats=False):
        _dev(xyz, residual, log_R, shift, scales, kscale)
        xyz, residual, log_R, shift = _f32c(xyz), _f32c(residual), _f32c(log_R), _f32c(shift)
        scales = _f32c(scales) if scales is not None else None
        ks = _f32c(kscale).reshape(1) if kscale is not None else None
        B, _, N = xyz.shape
        H = log_R.shape[0]
        assert xyz.shape[1] == 3 and residual.shape == (B, H * 3, N) and log_R.shape == (H, 3) and shift.shape == (H, 3)
        dev = xyz.device
        keys = torch.empty(B, H * dim, N, device=dev, dtype=torch.float32)
        lattice = torch.empty_like(keys)
        R = torch.empty(H, 3, 3, device=dev, dtype=torch.float32)
        lib = _lib.load()
        stats = ws = ticket = None
        ws_bytes = 0
        if with_stats:
            stats = torch.empty(2, device=dev, dtype=torch.float32)
            ws_bytes = lib.ct_lattice_fwd_workspace_bytes(B, H, N)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            tk = raster_tickets(dev, force=True)
            ticket = tk.data_ptr() + (tk.numel() - 1) * 4          # the buffer's last word (the raster kernels use the front)
        with _on(dev):
            _lib.check(lib.ct_lattice_so3_fwd(_ptr(xyz), _ptr(residual), _ptr(log_R), float(eps), _ptr(shift), _ptr(scales), _ptr(ks),
                                              _ptr(R), _ptr(keys), _ptr(lattice), _ptr(stats), _ptr(ws), ws_bytes, ticket, B, H, N, dim,
                                              _stream()), "ct_lattice_so3_fwd")
        ctx.save_for_backward(xyz, residual, log_R, R, shift, scales, ks, lattice)
        ctx.meta = (B, H, N, dim, kscale.shape if kscale is not None else None, float(eps))
        ctx.set_materialize_grads(False)
        if with_stats:
            ctx.mark_non_differentiable(stats)
            return keys, lattice, stats
        return keys, lattice

    @staticmethod
    def backward(ctx, g_keys, g_lattice, _g_stats=None):
        xyz, residual, log_R, R, shift, scales, ks, lattice = ctx.saved_tensors
        B, H, N, dim, ks_shape, eps = ctx.meta
        if g_keys is None and g_lattice is None:
            return (None,) * 9
        g_keys = _f32c(g_keys) if g_keys is not None else None
        g_lattice = _f32c(g_lattice) if g_lattice is not None else None
        g_xyz, g_res = torch.empty_like(xyz), torch.empty_like(residual)
        g_R, g_shift, g_log_R = torch.empty_like(R), torch.empty_like(shift), torch.empty_like(log_R)
        g_scales = torch.empty_like(scales) if scales is not None else None
        g_ks = torch.empty_like(ks) if ks is not None else None
        lib = _lib.load()
        with _on(xyz.device):
            ws_bytes = lib.ct_lattice_bwd_workspace_bytes(B, H, N)
            ws = torch.empty(ws_bytes, device=xyz.device, dtype=torch.uint8)
            _lib.check(lib.ct_lattice_so3_bwd(_ptr(xyz), _ptr(residual), _ptr(log_R), eps, _ptr(R), _ptr(shift), _ptr(scales), _ptr(ks),
                                              _ptr(lattice), _ptr(g_lattice), _ptr(g_keys), _ptr(g_xyz), _ptr(g_res), _ptr(g_log_R),
                                              _ptr(g_R), _ptr(g_shift), _ptr(g_scales), _ptr(g_ks), _ptr(ws), ws_bytes, B, H, N, dim,
                                              _stream()), "ct_lattice_so3_bwd")
        return (g_xyz, g_res, g_log_R, g_shift, g_scales, (g_ks.reshape(ks_shape) if g_ks is not None else None), None, None, None)


def lattice_so3(xyz, residual, log_R, shift, scales, kscale, dim, eps=1e-4, with_stats=False):
    """(keys, tanh(keys)[, key statistics]) of an MHCT block from the transformer's so3 parameters log_R [H,3]
    (layers/utils.py:25-34,53-61; eps: pytorch3d's so3_exponential_map default)."""
    return LatticeSo3Fn.apply(xyz, residual, log_R, shift, scales, kscale, dim, eps, with_stats)


class So3ExpFn(torch.autograd.Function):
    """so3 exponential map log_R [H,3] -> R [H,3,3] (pytorch3d's, layers/utils.py:6,29,56), one launch each way."""

    @staticmethod
    def forward(ctx, log_R, eps):
        _dev(log_R)
        log_R = _f32c(log_R)
        H = log_R.shape[0]
        R = torch.empty(H, 3, 3, device=log_R.device, dtype=torch.float32)
        lib = _lib.load()
        with _on(log_R.device):
            _lib.check(lib.ct_so3_exp_fwd(_ptr(log_R), _ptr(R), H, float(eps), _stream()), "ct_so3_exp_fwd")
        ctx.save_for_backward(log_R)
        ctx.eps = float(eps)
        return R

    @staticmethod
    def backward(ctx, g_R):
        (log_R,) = ctx.saved_tensors
        g_R = _f32c(g_R)
        g = torch.empty_like(log_R)
        lib = _lib.load()
        with _on(log_R.device):
            _lib.check(lib.ct_so3_exp_bwd(_ptr(log_R), _ptr(g_R), _ptr(g), log_R.shape[0], ctx.eps, _stream()), "ct_so3_exp_bwd")
        return g, None


def so3_exp(log_R, eps=1e-4):
    return So3ExpFn.apply(log_R, eps)


def _batch_stride(t, C, N):
    """Batch stride (floats) of t if it is a [B,C,N] tensor whose rows are contiguous, whose channels are N apart and
    whose batches are a multiple of 4 >= C*N apart, 16-byte aligned — a contiguous tensor or a channel slice of a wider
    one (what torch.cat's backward hands out); None otherwise."""
    if (t.dim() == 3 and t.dtype == torch.float32 and t.size(1) == C and t.size(2) == N and t.stride(2) == 1
            and (t.stride(1) == N or C == 1) and t.stride(0) >= C * N and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
        return t.stride(0)
    return None


class AdaInFn(torch.autograd.Function):
    """relu?(instance_norm(x) * (gamma + 1) + beta) [+ residual], gamma_beta [B,2,C] (layers/utils.py:88-97).  x may be a
    channel slice of a wider tensor (read where it lies), and so may the cotangent."""

    @staticmethod
    def forward(ctx, x, gamma_beta, eps, relu, residual):
        _dev(x, gamma_beta)
        if x.dtype != torch.float32:
            raise TypeError("expected float32, got %s" % x.dtype)
        B, C, N = x.shape
        xbs = _batch_stride(x, C, N)
        if xbs is None:
            x, xbs = x.contiguous(), 0
        assert gamma_beta.shape == (B, 2, C)
        gamma_beta, gbbs = _gb_arg(gamma_beta, B, C)
        rbs = 0
        if residual is not None:
            rbs = _batch_stride(residual, C, N)
            if rbs is None:
                residual, rbs = _f32c(residual), 0
        y = torch.empty(B, C, N, device=x.device, dtype=torch.float32)
        mean = torch.empty(B * C, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        slots = _amax_slots(B * C, x.device)
        with _on(x.device):
            _adain_group_fwd([dict(x=_ptr(x), xbs=xbs, gb=gamma_beta, gbbs=gbbs, res=_ptr(residual), rbs=rbs, y=_ptr(y), ybs=0, mean=mean,
                                   rstd=rstd, amax=_ptr(slots), abs=0, C=C, eps=eps, relu=bool(relu))], B, N)
        tag_amax(y, slots)
        ctx.save_for_backward(x, gamma_beta, mean, rstd)
        ctx.relu = int(bool(relu))
        ctx.xbs = xbs
        ctx.gbbs = gbbs
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma_beta, mean, rstd = ctx.saved_tensors
        B, C, N = x.shape
        gybs = _batch_stride(gy, C, N)
        if gybs is None:
            gy, gybs = _f32c(gy), 0
        gx = torch.empty(B, C, N, device=x.device, dtype=torch.float32)
        g_gb = torch.empty(B, 2, C, device=x.device, dtype=torch.float32)
        slots = _amax_slots(B * C, x.device)
        with _on(x.device):
            _adain_group_bwd([dict(x=_ptr(x), xbs=ctx.xbs, gb=gamma_beta, gbbs=ctx.gbbs, mean=mean, rstd=rstd, gy=_ptr(gy), gybs=gybs,
                                   gx=_ptr(gx), gxbs=0, g_gb=g_gb, amax=_ptr(slots), abs=0, C=C, relu=ctx.relu)], B, N)
        tag_amax(gx, slots)
        return gx, g_gb, None, None, (gy if ctx.has_residual else None)


def adain(x, gamma_beta, eps=1e-5, relu=False, residual=None):
    """Adaptive instance norm of x [B,C,N] with per-(b,c) scale (+1) and bias gamma_beta [B,2,C]; `residual` is added
    to the result in the same pass."""
    return AdaInFn.apply(x, gamma_beta, eps, relu, residual)


class StyleProjFn(torch.autograd.Function):
    """The style projections of ALL adaptive instance norms of a block — `linear_i(style)` for every AdaIn1dUpd
    (layers/utils.py:90; layers/multihead_ct_adain.py: keys_bn / values_bn of each head, the heads' `after` norms, the union's `after`
    and shortcut norms) — as ONE product with the weights stacked: seven [B,L] x [L,2C_i] library GEMMs of 8-15 us, their 21
    backward products and the six accumulations of the style cotangent become 3 + 4 launches.  Arguments: style [B,L], then
    (weight [2C_i,L], bias [2C_i]) per norm.  Returns one [B,2,C_i] VIEW per norm into the stacked result (strides (sum, C_i, 1):
    the AdaIN kernels read it where it lies, `_gb_arg`)."""

    @staticmethod
    def forward(ctx, style, *wb):
        ws, bs = wb[0::2], wb[1::2]
        W = torch.cat(ws, dim=0)
        out = torch.addmm(torch.cat(bs, dim=0), style, W.t())              # [B, sum 2C_i]
        ctx.save_for_backward(style, W)
        ctx.sizes = [w.size(0) for w in ws]
        outs, o = [], 0
        for d in ctx.sizes:
            outs.append(out[:, o:o + d].unflatten(1, (2, d // 2)))
            o += d
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        style, W = ctx.saved_tensors
        B = style.size(0)
        parts = [(g.reshape(B, d) if g is not None else style.new_zeros(B, d)) for g, d in zip(gouts, ctx.sizes)]
        g = torch.cat(parts, dim=1)                                           # [B, sum 2C_i]
        g_style = g @ W if ctx.needs_input_grad[0] else None
        g_W, g_b = g.t() @ style, g.sum(0)
        grads, o = [g_style], 0
        for d in ctx.sizes:                       # row ranges of the stacked gradients: dense views, no copies
            grads += [g_W[o:o + d], g_b[o:o + d]]
            o += d
        return tuple(grads)


def _gb_arg(gb, B, C):
    """gamma_beta [B,2,C] for the AdaIN kernels without a copy: (tensor, batch stride in floats, 0 = contiguous).  A column
    range of a stacked style projection [B, sum 2*C_i] viewed as [B,2,C] has strides (sum, C, 1): the kernels take the sum."""
    if gb.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % gb.dtype)
    if gb.dim() == 3 and gb.shape == (B, 2, C) and gb.stride(2) == 1 and gb.stride(1) == C and (B == 1 or gb.stride(0) >= 2 * C):
        return gb, (0 if B == 1 or gb.stride(0) == 2 * C else gb.stride(0))
    return gb.contiguous(), 0


def _adain_group_fwd(items, B, N):
    """ct_adain_fwd_amax of every item — in ONE launch when there are several (ct_adain_group_fwd).  items: dicts with x (ptr),
    xbs, gb, res, rbs, y (ptr), ybs, mean, rstd, amax (ptr or None), abs (amax batch stride), C, eps, relu."""
    lib = _lib.load()
    if B == 0 or N == 0:
        return
    items = [it for it in items if it["C"] > 0]
    step = _lib.BN_GROUP_MAX if BN_GROUP_LAUNCH else 1
    for i0 in range(0, len(items), step):
        chunk = items[i0:i0 + step]
        arr = (_lib.AdainFwdItem * len(chunk))()
        for e, it in zip(arr, chunk):
            e.x, e.x_batch_stride, e.gamma_beta, e.residual, e.residual_batch_stride = it["x"], it["xbs"], _ptr(it["gb"]), it["res"], it["rbs"]
            e.y, e.y_batch_stride, e.mean, e.rstd = it["y"], it["ybs"], _ptr(it["mean"]), _ptr(it["rstd"])
            e.amax_out, e.amax_batch_stride, e.C, e.eps, e.relu = it["amax"], it["abs"], it["C"], float(it["eps"]), int(it["relu"])
            e.gamma_beta_batch_stride = it.get("gbbs", 0)
        _lib.check(lib.ct_adain_group_fwd(ctypes.addressof(arr), len(chunk), B, N, _stream()), "ct_adain_group_fwd")


def _adain_group_bwd(items, B, N):
    """ct_adain_bwd_amax of every item, in ONE launch when there are several.  items: dicts with x (ptr), xbs, gb, mean, rstd,
    gy (ptr), gybs, gx (ptr), gxbs, g_gb, amax, abs, C, relu."""
    lib = _lib.load()
    if B == 0:
        return
    if N == 0:
        for it in items:
            it["g_gb"].zero_()
        return
    items = [it for it in items if it["C"] > 0]
    step = _lib.BN_GROUP_MAX if BN_GROUP_LAUNCH else 1
    for i0 in range(0, len(items), step):
        chunk = items[i0:i0 + step]
        arr = (_lib.AdainBwdItem * len(chunk))()
        for e, it in zip(arr, chunk):
            e.x, e.x_batch_stride, e.gamma_beta, e.mean, e.rstd = it["x"], it["xbs"], _ptr(it["gb"]), _ptr(it["mean"]), _ptr(it["rstd"])
            e.gy, e.gy_batch_stride, e.gx, e.gx_batch_stride, e.g_gamma_beta = it["gy"], it["gybs"], it["gx"], it["gxbs"], _ptr(it["g_gb"])
            e.amax_out, e.amax_batch_stride, e.C, e.relu = it["amax"], it["abs"], it["C"], int(it["relu"])
            e.gamma_beta_batch_stride = it.get("gbbs", 0)
        _lib.check(lib.ct_adain_group_bwd(ctypes.addressof(arr), len(chunk), B, N, _stream()), "ct_adain_group_bwd")


class UnionKeysValuesAdaInFn(torch.autograd.Function):
    """UnionKeysValuesFn for the AdaIN blocks (layers/multihead_ct_adain.py:104-111): one stacked GEMM for the heads'
    keys_values_pred projections, keys_bn / values_bn = adaptive instance norms on channel ranges of its output.
    Arguments: n, x, eps, then per head: weight [Co,Cin,1], gamma_beta of keys_bn [B,2,Ck], gamma_beta of values_bn
    [B,2,Cv].  Returns (keys_res_0, values_0, keys_res_1, values_1, ...)."""

    @staticmethod
    def forward(ctx, n, x, eps, *args):
        ctx.passthrough = n < 0          # x itself as the first output: see UnionKeysValuesFn
        n = abs(n)
        heads = [args[i * 3:(i + 1) * 3] for i in range(n)]
        x_in = x
        x = _f32c(x)
        _dev(x)
        B, Cin, N = x.shape
        Wc = torch.cat([h[0][:, :, 0] for h in heads], dim=0)
        Ct = Wc.size(0)
        y, am_w, am_x, Wt = pw_forward(Wc, x, ctx.needs_input_grad[1])
        outs, saved, meta, items, c0 = [], [], [], [], 0
        for h in heads:
            for gb in h[1:3]:
                C = gb.size(2)
                gb, gbbs = _gb_arg(gb, B, C)
                o = torch.empty(B, C, N, device=x.device, dtype=torch.float32)
                mean = torch.empty(B * C, device=x.device, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                items.append(dict(x=_ptr(y) + c0 * N * 4, xbs=Ct * N, gb=gb, gbbs=gbbs, res=None, rbs=0, y=_ptr(o), ybs=0, mean=mean,
                                  rstd=rstd, amax=None, abs=0, C=C, eps=eps, relu=0))
                outs.append(o)
                saved += [gb, mean, rstd]
                meta.append((c0, C))
                c0 += C
        with _on(x.device):
            _adain_group_fwd(items, B, N)
        assert c0 == Ct, "keys_bn + values_bn must cover the projections"
        ctx.save_for_backward(x, y, Wc, *saved)
        ctx.am = (am_w, am_x, Wt)
        ctx.meta = meta
        ctx.couts = [h[0].size(0) for h in heads]
        if ctx.passthrough:
            return (x_in,) + tuple(outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        g_skip = None
        if ctx.passthrough:
            g_skip, gouts = gouts[0], gouts[1:]
        x, y, Wc = ctx.saved_tensors[:3]
        saved = ctx.saved_tensors[3:]
        B, Cin, N = x.shape
        Ct = Wc.size(0)
        g_y = torch.empty_like(y)
        g_gbs, items, keep = [], [], []
        slots = _amax_slots(B * Ct, x.device)           # [B][Ct]: every norm writes its channel range of every cloud
        for i, (c0, C) in enumerate(ctx.meta):
            gb, mean, rstd = saved[i * 3:(i + 1) * 3]
            gy = gouts[i]
            if gy is None:
                gy = torch.zeros(B, C, N, device=x.device, dtype=torch.float32)
            gybs = _batch_stride(gy, C, N)
            if gybs is None:
                gy, gybs = _f32c(gy), 0
            keep.append(gy)
            g_gb = torch.empty(B, 2, C, device=x.device, dtype=torch.float32)
            items.append(dict(x=_ptr(y) + c0 * N * 4, xbs=Ct * N, gb=gb, gbbs=_gb_arg(gb, B, C)[1], mean=mean, rstd=rstd, gy=_ptr(gy), gybs=gybs,
                              gx=_ptr(g_y) + c0 * N * 4, gxbs=Ct * N, g_gb=g_gb,
                              amax=None if slots is None else _ptr(slots) + 4 * c0, abs=Ct, C=C, relu=0))
            g_gbs.append(g_gb)
        with _on(x.device):
            _adain_group_bwd(items, B, N)
        g_x, g_Wc = pw_backward(Wc, x, g_y, ctx.am[0], ctx.am[1], ctx.needs_input_grad[1], True,
                                am_g=None if slots is None else slots.view(-1, Ct), Wt=ctx.am[2],
                                add_gx=g_skip if ctx.needs_input_grad[1] else None)
        grads, r0 = [None, g_x, None], 0
        for hi, Co in enumerate(ctx.couts):
            grads += [g_Wc[r0:r0 + Co].unsqueeze(-1), g_gbs[2 * hi], g_gbs[2 * hi + 1]]
            r0 += Co
        return tuple(grads)


class JoinAdaInReluFn(torch.autograd.Function):
    """cat([relu(adain_i(x_i)) for i], dim=1): the AdaIN heads' `after` stacks followed by the union's concatenation
    (layers/multihead_ct_adain.py:64-66,205-214) — JoinBnReluFn's counterpart.  Arguments: n, eps, then per head (x, gamma_beta)."""

    @staticmethod
    def forward(ctx, n, eps, *args):
        xs = [_f32c(args[2 * i]) for i in range(n)]
        gbs = [_gb_arg(args[2 * i + 1], xs[i].size(0), xs[i].size(1))[0] for i in range(n)]
        _dev(*xs)
        B, _, N = xs[0].shape
        Ct = sum(x.size(1) for x in xs)
        y = torch.empty(B, Ct, N, device=xs[0].device, dtype=torch.float32)
        slots = _amax_slots(B * Ct, y.device)
        saved, items, c0 = [], [], 0
        for x, gb in zip(xs, gbs):
            C = x.size(1)
            mean = torch.empty(B * C, device=y.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            items.append(dict(x=_ptr(x), xbs=0, gb=gb, gbbs=_gb_arg(gb, B, C)[1], res=None, rbs=0, y=_ptr(y) + c0 * N * 4, ybs=Ct * N,
                              mean=mean, rstd=rstd, amax=None if slots is None else _ptr(slots) + 4 * c0, abs=Ct, C=C, eps=eps, relu=1))
            saved += [x, gb, mean, rstd]
            c0 += C
        with _on(y.device):
            _adain_group_fwd(items, B, N)
        tag_amax(y, slots)
        ctx.save_for_backward(*saved)
        ctx.n = n
        return y

    @staticmethod
    def backward(ctx, gy):
        n, saved = ctx.n, ctx.saved_tensors
        B, Ct, N = gy.shape
        gybs = _batch_stride(gy, Ct, N)
        if gybs is None:
            gy, gybs = _f32c(gy), Ct * N
        grads, items, c0 = [None, None], [], 0
        for i in range(n):
            x, gb, mean, rstd = saved[i * 4:(i + 1) * 4]
            C = x.size(1)
            gx, g_gb = torch.empty_like(x), torch.empty(B, 2, C, device=x.device, dtype=torch.float32)
            slots = _amax_slots(B * C, gx.device)
            items.append(dict(x=_ptr(x), xbs=0, gb=gb, gbbs=_gb_arg(gb, B, C)[1], mean=mean, rstd=rstd, gy=_ptr(gy) + c0 * N * 4, gybs=gybs,
                              gx=_ptr(gx), gxbs=0,
                              g_gb=g_gb, amax=_ptr(slots), abs=0, C=C, relu=1))
            grads += [tag_amax(gx, slots), g_gb]
            c0 += C
        with _on(gy.device):
            _adain_group_bwd(items, B, N)
        return tuple(grads)


# ---------------------------------------------------------------------------
# training-mode BatchNorm1d (+ ReLU, + skip) groups, with or without an exchange of statistics between ranks
# ---------------------------------------------------------------------------
_sync_stats_collectives = 0
BN_GROUP_LAUNCH = os.environ.get("CLOUDCT_BN_GROUP", "1") != "0"     # A/B switch: the norms of a group one by one


def sync_stats_collectives():
    """Number of statistics collectives (one all_gather per norm group in forward, one all_reduce in backward) issued
    by this process so far — bench.py reports the per-step figure."""
    return _sync_stats_collectives


SYNC_STATS_FORCE = os.environ.get("CLOUDCT_SYNCBN_FORCE", "0") == "1"


def _sync_group(bn):
    """The process group a norm module exchanges its batch statistics over, or None: a plain BatchNorm1d, no process
    group initialised, or a SyncBatchNorm whose group has a single rank (torch's own SyncBatchNorm then runs the plain
    batch norm too: nn/modules/batchnorm.py `need_sync`)."""
    if not isinstance(bn, torch.nn.SyncBatchNorm):
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    pg = bn.process_group if bn.process_group is not None else dist.group.WORLD
    # (SYNC_STATS_FORCE: keep the exchange path on in a one-rank group too — what a multi-rank step enqueues, collectives
    #  included, then runs and is timed on a single GPU: bench.py --mode ddp-step, tests/test_ddp_gpu.py)
    return pg if dist.get_world_size(pg) > 1 or SYNC_STATS_FORCE else None


def norms_share_group(bns):
    """True when the norms of a fused group are of ONE type and exchange their statistics over ONE process group: the
    group kernels take both from the first norm, so a mix (an unconverted BatchNorm1d beside SyncBatchNorms, or two
    process groups) must go through the modules one by one."""
    first, group = type(bns[0]), _sync_group(bns[0])
    return all(type(bn) is first and _sync_group(bn) is group for bn in bns[1:])


def _bn_group_fwd(items, B, N, device, group):
    """Run the norms of one group.  items: dicts with x (data_ptr), xbs, C, w, b, rm, rv, nbt, eps, mom, relu, res (ptr or
    None), rbs, y (ptr), ybs and, optionally, amax (ptr to C floats: the per-channel max |y| as written).  Returns ([(mean, rstd)] per item, count) — count is None without a group, else a 1-float
    device tensor with the job's values per channel.  With a group: local statistics of ALL items into one buffer, ONE
    all_gather, then the normalising kernels merge the ranks' statistics themselves (csrc/ct_bnorm.hip mode 1 / 2)."""
    global _sync_stats_collectives
    lib = _lib.load()
    stats = []
    if group is None and 1 < len(items) <= _lib.BN_GROUP_MAX and BN_GROUP_LAUNCH:
        # the norms of a block's group in ONE launch (a workgroup per channel of every norm)
        arr = (_lib.BnFwdItem * len(items))()
        for e, it in zip(arr, items):
            mean = torch.empty(it["C"], device=device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            e.x, e.x_batch_stride, e.weight, e.bias = it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"])
            e.running_mean, e.running_var, e.num_batches_tracked = _ptr(it["rm"]), _ptr(it["rv"]), _ptr(it["nbt"])
            e.residual, e.residual_batch_stride, e.y, e.y_batch_stride = it["res"], it["rbs"], it["y"], it["ybs"]
            e.save_mean, e.save_rstd, e.amax_out = _ptr(mean), _ptr(rstd), it.get("amax")
            e.C, e.eps, e.momentum, e.relu = it["C"], float(it["eps"]), float(it["mom"]), int(it["relu"])
            stats.append((mean, rstd))
        _lib.check(lib.ct_bn_group_fwd(ctypes.addressof(arr), len(items), B, N, _stream()), "ct_bn_group_fwd")
        return stats, None
    if group is None:
        for it in items:
            mean = torch.empty(it["C"], device=device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            _lib.check(lib.ct_bn_relu_fwd_amax(it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"]), _ptr(it["rm"]), _ptr(it["rv"]),
                                               _ptr(it["nbt"]), it["res"], it["rbs"], it["y"], it["ybs"], _ptr(mean), _ptr(rstd),
                                               it.get("amax"), B, it["C"], N, float(it["eps"]), float(it["mom"]),
                                               int(it["relu"]), _stream()), "ct_bn_relu_fwd")
            stats.append((mean, rstd))
        return stats, None
    import torch.distributed as dist
    world = dist.get_world_size(group)
    Ct = sum(it["C"] for it in items)
    stride = 2 * Ct + 1
    local = torch.empty(stride, device=device, dtype=torch.float32)
    if len(items) <= _lib.BN_GROUP_MAX and BN_GROUP_LAUNCH:
        # every phase of the group in ONE launch: statistics -> all_gather -> normalise (2 launches + 1 collective per group)
        arr = (_lib.BnFwdItem * len(items))()
        outs = []
        for e, it in zip(arr, items):
            mean = torch.empty(it["C"], device=device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            e.x, e.x_batch_stride, e.weight, e.bias = it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"])
            e.running_mean, e.running_var, e.num_batches_tracked = _ptr(it["rm"]), _ptr(it["rv"]), _ptr(it["nbt"])
            e.residual, e.residual_batch_stride, e.y, e.y_batch_stride = it["res"], it["rbs"], it["y"], it["ybs"]
            e.save_mean, e.save_rstd, e.amax_out = _ptr(mean), _ptr(rstd), it.get("amax")
            e.C, e.eps, e.momentum, e.relu = it["C"], float(it["eps"]), float(it["mom"]), int(it["relu"])
            outs.append((mean, rstd))
        _lib.check(lib.ct_bn_group_stats_fwd(ctypes.addressof(arr), len(items), B, N, _ptr(local), _stream()), "ct_bn_group_stats_fwd")
        gathered = torch.empty(world * stride, device=device, dtype=torch.float32)
        work = dist.all_gather_into_tensor(gathered, local, group=group, async_op=True)
        _sync_stats_collectives += 1
        count = torch.empty(1, device=device, dtype=torch.float32)
        work.wait()                   # stream-level wait (no host synchronisation with the RCCL backend)
        _lib.check(lib.ct_bn_group_apply_fwd(ctypes.addressof(arr), len(items), B, N, _ptr(gathered), world, _ptr(count), _stream()),
                   "ct_bn_group_apply_fwd")
        return outs, count
    base, c0 = local.data_ptr(), 0
    for i, it in enumerate(items):
        _lib.check(lib.ct_bn_stats_fwd(it["x"], it["xbs"], base + 4 * c0, base + 4 * (Ct + c0),
                                       base + 4 * (stride - 1) if i == 0 else None, B, it["C"], N, _stream()), "ct_bn_stats_fwd")
        c0 += it["C"]
    gathered = torch.empty(world * stride, device=device, dtype=torch.float32)
    # non-blocking: the collective is enqueued on RCCL's own stream (behind the statistics kernels) and the compute stream
    # waits for it only where the first normalising kernel is enqueued — the host prepares those launches meanwhile, and
    # whatever else is already queued on other streams (DDP's gradient buckets in backward) is not held up behind it
    work = dist.all_gather_into_tensor(gathered, local, group=group, async_op=True)
    _sync_stats_collectives += 1
    count = torch.empty(1, device=device, dtype=torch.float32)
    outs = [(torch.empty(it["C"], device=device, dtype=torch.float32), torch.empty(it["C"], device=device, dtype=torch.float32))
            for it in items]
    work.wait()                       # stream-level wait (no host synchronisation with the RCCL backend)
    gb, c0 = gathered.data_ptr(), 0
    for i, it in enumerate(items):
        mean, rstd = outs[i]
        _lib.check(lib.ct_bn_apply_fwd_amax(it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"]), gb + 4 * c0, gb + 4 * (Ct + c0),
                                            gb + 4 * (stride - 1), world, stride, _ptr(it["rm"]), _ptr(it["rv"]), _ptr(it["nbt"]),
                                            it["res"], it["rbs"], it["y"], it["ybs"], _ptr(mean), _ptr(rstd),
                                            _ptr(count) if i == 0 else None, it.get("amax"), B, it["C"], N, float(it["eps"]),
                                            float(it["mom"]), int(it["relu"]), _stream()), "ct_bn_apply_fwd")
        stats.append((mean, rstd))
        c0 += it["C"]
    # (`gathered` is read by the kernels just enqueued: torch's caching allocator keeps it alive on this stream)
    return stats, count


def _bn_group_bwd(items, B, N, device, group, count):
    """Backward of a norm group.  items: dicts with x, xbs, C, w, b, mean, rstd, gy (ptr), gybs, gx (ptr), gxbs, relu.
    Returns [(g_weight, g_bias)] per item (this rank's sums: DDP averages parameter gradients itself).  With a group:
    the two per-channel sums of ALL items into one buffer, ONE all_reduce, then the input-gradient kernels."""
    global _sync_stats_collectives
    lib = _lib.load()
    out = []
    if group is None and 1 < len(items) <= _lib.BN_GROUP_MAX and BN_GROUP_LAUNCH:
        arr = (_lib.BnBwdItem * len(items))()
        for e, it in zip(arr, items):
            g_w = torch.empty(it["C"], device=device, dtype=torch.float32)
            g_b = torch.empty_like(g_w)
            e.x, e.x_batch_stride, e.weight, e.bias = it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"])
            e.save_mean, e.save_rstd, e.gy, e.gy_batch_stride = _ptr(it["mean"]), _ptr(it["rstd"]), it["gy"], it["gybs"]
            e.gx, e.gx_batch_stride, e.g_weight, e.g_bias, e.amax_out = it["gx"], it["gxbs"], _ptr(g_w), _ptr(g_b), it.get("amax")
            e.C, e.relu = it["C"], int(it["relu"])
            out.append((g_w, g_b))
        _lib.check(lib.ct_bn_group_bwd(ctypes.addressof(arr), len(items), B, N, _stream()), "ct_bn_group_bwd")
        return out
    if group is None:
        for it in items:
            g_w = torch.empty(it["C"], device=device, dtype=torch.float32)
            g_b = torch.empty_like(g_w)
            _lib.check(lib.ct_bn_relu_bwd_amax(it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"]), _ptr(it["mean"]), _ptr(it["rstd"]),
                                               it["gy"], it["gybs"], it["gx"], it["gxbs"], _ptr(g_w), _ptr(g_b), it.get("amax"),
                                               B, it["C"], N, int(it["relu"]), _stream()), "ct_bn_relu_bwd")
            out.append((g_w, g_b))
        return out
    import torch.distributed as dist
    Ct = sum(it["C"] for it in items)
    sums = torch.empty(2 * Ct, device=device, dtype=torch.float32)       # [sum g' | sum g' * xhat]
    if len(items) <= _lib.BN_GROUP_MAX and BN_GROUP_LAUNCH:
        arr = (_lib.BnBwdItem * len(items))()
        for e, it in zip(arr, items):
            e.x, e.x_batch_stride, e.weight, e.bias = it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"])
            e.save_mean, e.save_rstd, e.gy, e.gy_batch_stride = _ptr(it["mean"]), _ptr(it["rstd"]), it["gy"], it["gybs"]
            e.gx, e.gx_batch_stride, e.g_weight, e.g_bias, e.amax_out = it["gx"], it["gxbs"], None, None, it.get("amax")
            e.C, e.relu = it["C"], int(it["relu"])
        local = torch.empty_like(sums)                                    # this rank's g_bias / g_weight: written by the same launch
        _lib.check(lib.ct_bn_group_reduce_bwd_copy(ctypes.addressof(arr), len(items), B, N, _ptr(sums), _ptr(local), _stream()),
                   "ct_bn_group_reduce_bwd_copy")
        work = dist.all_reduce(sums, group=group, async_op=True)
        _sync_stats_collectives += 1
        work.wait()
        _lib.check(lib.ct_bn_group_apply_bwd(ctypes.addressof(arr), len(items), B, N, _ptr(sums), _ptr(count), _stream()),
                   "ct_bn_group_apply_bwd")
        c0 = 0
        for it in items:
            out.append((local[Ct + c0:Ct + c0 + it["C"]], local[c0:c0 + it["C"]]))
            c0 += it["C"]
        return out
    sb, c0 = sums.data_ptr(), 0
    for it in items:
        _lib.check(lib.ct_bn_reduce_bwd(it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"]), _ptr(it["mean"]), _ptr(it["rstd"]),
                                        it["gy"], it["gybs"], sb + 4 * c0, sb + 4 * (Ct + c0), B, it["C"], N, int(it["relu"]),
                                        _stream()), "ct_bn_reduce_bwd")
        c0 += it["C"]
    local = sums.clone()                                                  # this rank's g_bias / g_weight
    work = dist.all_reduce(sums, group=group, async_op=True)             # non-blocking, as in _bn_group_fwd
    _sync_stats_collectives += 1
    work.wait()
    c0 = 0
    for it in items:
        _lib.check(lib.ct_bn_apply_bwd_amax(it["x"], it["xbs"], _ptr(it["w"]), _ptr(it["b"]), _ptr(it["mean"]), _ptr(it["rstd"]),
                                            it["gy"], it["gybs"], sb + 4 * c0, sb + 4 * (Ct + c0), _ptr(count), it["gx"], it["gxbs"],
                                            it.get("amax"), B, it["C"], N, int(it["relu"]), _stream()), "ct_bn_apply_bwd")
        out.append((local[Ct + c0:Ct + c0 + it["C"]], local[c0:c0 + it["C"]]))
        c0 += it["C"]
    return out


class BnReluFn(torch.autograd.Function):
    """relu?(batch_norm(x)) [+ residual] in training mode (batch statistics; running statistics and num_batches_tracked
    updated in place by the kernel): the nn.Sequential(BatchNorm1d, ReLU) tail of the blocks' `after` stacks and the
    union's skip connection (layers/multihead_ct.py:67-68,149-153,198).  `group`: the process group of a SyncBatchNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, eps, momentum, relu, residual, group=None):
        _dev(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        B, C, N = x.shape
        y = torch.empty_like(x)
        rbs = 0
        if residual is not None:
            rbs = _batch_stride(residual, C, N)
            if rbs is None:
                residual = _f32c(residual)
                rbs = 0
        slots = _amax_slots(C, x.device)
        with _on(x.device):
            stats, count = _bn_group_fwd([dict(x=_ptr(x), xbs=0, C=C, w=weight, b=bias, rm=running_mean, rv=running_var, nbt=nbt,
                                               eps=eps, mom=momentum, relu=relu, res=_ptr(residual), rbs=rbs, y=_ptr(y), ybs=0,
                                               amax=_ptr(slots))],
                                         B, N, x.device, group)
        tag_amax(y, slots)
        mean, rstd = stats[0]
        ctx.save_for_backward(x, weight, bias, mean, rstd, count)
        ctx.relu = int(bool(relu))
        ctx.has_residual = residual is not None
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, rstd, count = ctx.saved_tensors
        B, C, N = x.shape
        gybs = _batch_stride(gy, C, N)          # a slice of the concatenation's cotangent is read where it lies
        if gybs is None:
            gy = _f32c(gy)
            gybs = 0
        gx = torch.empty_like(x)
        slots = _amax_slots(C, x.device)
        with _on(x.device):
            ((g_w, g_b),) = _bn_group_bwd([dict(x=_ptr(x), xbs=0, C=C, w=weight, b=bias, mean=mean, rstd=rstd, gy=_ptr(gy),
                                                gybs=gybs, gx=_ptr(gx), gxbs=0, relu=ctx.relu, amax=_ptr(slots))],
                                          B, N, x.device, ctx.group, count)
        tag_amax(gx, slots)
        return gx, g_w, g_b, None, None, None, None, None, None, (gy if ctx.has_residual else None), None


class SplitBnFn(torch.autograd.Function):
    """(key_bn(x[:, :Ck]), values_bn(x[:, Ck:])) in training mode — the two BatchNorms on the halves of the
    keys_values_pred output (layers/multihead_ct.py:89-91).  The kernels read the channel slices of x where they lie
    (batch stride C*N) and the backward writes both input cotangents straight into the halves of ONE [B,C,N] tensor:
    neither the contiguous copies of the slices nor the concatenation of their cotangents exist."""

    @staticmethod
    def forward(ctx, x, wk, bk, rmk, rvk, nbk, epsk, momk, wv, bv, rmv, rvv, nbv, epsv, momv, group=None):
        _dev(x, wk, wv)
        x = _f32c(x)
        B, C, N = x.shape
        Ck = wk.numel()
        Cv = C - Ck
        assert wv.numel() == Cv
        items, outs = [], []
        for c0, Cs, w, b, rm, rv, nb, eps, mom in ((0, Ck, wk, bk, rmk, rvk, nbk, epsk, momk),
                                                   (Ck, Cv, wv, bv, rmv, rvv, nbv, epsv, momv)):
            y = torch.empty(B, Cs, N, device=x.device, dtype=torch.float32)
            items.append(dict(x=_ptr(x) + c0 * N * 4, xbs=C * N, C=Cs, w=_f32c(w), b=_f32c(b), rm=rm, rv=rv, nbt=nb, eps=eps,
                              mom=mom, relu=0, res=None, rbs=0, y=_ptr(y), ybs=0))
            outs.append(y)
        with _on(x.device):
            stats, count = _bn_group_fwd(items, B, N, x.device, group)
        saved = []
        for it, (mean, rstd) in zip(items, stats):
            saved += [it["w"], it["b"], mean, rstd]
        ctx.save_for_backward(x, count, *saved)
        ctx.Ck = Ck
        ctx.group = group
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, gk, gv):
        x, count, wk, bk, mk, rk, wv, bv, mv, rv = ctx.saved_tensors
        B, C, N = x.shape
        Ck = ctx.Ck
        gx = torch.empty_like(x)
        items = []
        for c0, Cs, w, b, mean, rstd, gy in ((0, Ck, wk, bk, mk, rk, gk), (Ck, C - Ck, wv, bv, mv, rv, gv)):
            if gy is None:
                gy = torch.zeros(B, Cs, N, device=x.device, dtype=torch.float32)
            gybs = _batch_stride(gy, Cs, N)
            if gybs is None:
                gy, gybs = _f32c(gy), 0
            items.append(dict(x=_ptr(x) + c0 * N * 4, xbs=C * N, C=Cs, w=w, b=b, mean=mean, rstd=rstd, gy=_ptr(gy), gybs=gybs,
                              gx=_ptr(gx) + c0 * N * 4, gxbs=C * N, relu=0, keep=gy))
        with _on(x.device):
            (gwk, gbk), (gwv, gbv) = _bn_group_bwd(items, B, N, x.device, ctx.group, count)
        return gx, gwk, gbk, None, None, None, None, None, gwv, gbv, None, None, None, None, None, None


class JoinBnReluFn(torch.autograd.Function):
    """cat([relu(bn_i(x_i)) for i], dim=1) — the heads' `after` stacks of a union block followed by its concatenation
    (layers/multihead_ct.py:67-68,187-196): every head's kernel writes its channel range of the result where it belongs
    and, in backward, reads its range of the cotangent where it lies; the separate outputs and their copy never exist.
    Arguments: n, group, then per head (x, weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum)."""

    @staticmethod
    def forward(ctx, n, group, *args):
        heads = [args[i * 8:(i + 1) * 8] for i in range(n)]
        xs = [_f32c(h[0]) for h in heads]
        _dev(*xs)
        B, _, N = xs[0].shape
        Ct = sum(x.size(1) for x in xs)
        y = torch.empty(B, Ct, N, device=xs[0].device, dtype=torch.float32)
        slots = _amax_slots(Ct, y.device)
        items, c0 = [], 0
        for x, (_, w, b, rm, rv, nbt, eps, mom) in zip(xs, heads):
            C = x.size(1)
            items.append(dict(x=_ptr(x), xbs=0, C=C, w=_f32c(w), b=_f32c(b), rm=rm, rv=rv, nbt=nbt, eps=eps, mom=mom, relu=1,
                              res=None, rbs=0, y=_ptr(y) + c0 * N * 4, ybs=Ct * N,
                              amax=None if slots is None else _ptr(slots) + 4 * c0))
            c0 += C
        with _on(y.device):
            stats, count = _bn_group_fwd(items, B, N, y.device, group)
        tag_amax(y, slots)
        saved = []
        for x, it, (mean, rstd) in zip(xs, items, stats):
            saved += [x, it["w"], it["b"], mean, rstd]
        ctx.save_for_backward(count, *saved)
        ctx.n = n
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n
        count = ctx.saved_tensors[0]
        saved = ctx.saved_tensors[1:]
        B, Ct, N = gy.shape
        gybs = _batch_stride(gy, Ct, N)
        if gybs is None:
            gy, gybs = _f32c(gy), Ct * N
        items, gxs, c0 = [], [], 0
        for i in range(n):
            x, w, b, mean, rstd = saved[i * 5:(i + 1) * 5]
            C = x.size(1)
            gx = torch.empty_like(x)
            gxs.append(gx)
            slots = _amax_slots(C, gx.device)
            items.append(dict(x=_ptr(x), xbs=0, C=C, w=w, b=b, mean=mean, rstd=rstd, gy=_ptr(gy) + c0 * N * 4, gybs=gybs,
                              gx=_ptr(gx), gxbs=0, relu=1, amax=_ptr(slots), slots=slots))
            c0 += C
        with _on(gy.device):
            wb = _bn_group_bwd(items, B, N, gy.device, ctx.group, count)
        for gx, it in zip(gxs, items):
            tag_amax(gx, it["slots"])
        grads = [None, None]
        for gx, (g_w, g_b) in zip(gxs, wb):
            grads += [gx, g_w, g_b, None, None, None, None, None]
        return tuple(grads)


def join_bn_relu(xs, bns):
    """cat([relu(bn(x)) for x, bn in zip(xs, bns)], dim=1) through JoinBnReluFn; the caller checked bn_relu_eligible for
    every pair."""
    args = []
    for x, bn in zip(xs, bns):
        args += [x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum]
    return JoinBnReluFn.apply(len(xs), _sync_group(bns[0]), *args)


class UnionKeysValuesFn(torch.autograd.Function):
    """The `keys_values_pred` projections of all heads of a union block and the key_bn / values_bn norms on their
    outputs (layers/multihead_ct.py:31-33,89-91), as ONE GEMM with the heads' weights stacked: the heads share the input,
    so forward is one [sum Co, Cin] x [Cin, N] product per cloud instead of one per head, the data gradient one product
    with K = sum Co instead of one per head plus autograd's accumulation of their results, the weight gradient one
    product.  The norms read their channel ranges of the GEMM output where they lie and write their input cotangents
    into the ranges of ONE tensor, which is the data / weight gradient GEMMs' operand.  Under SyncBatchNorm the 2n norms
    share ONE statistics all_gather in forward and ONE all_reduce in backward.
    Arguments: n, group, x, then per head: weight [Co,Cin,1], key_bn (w, b, rm, rv, nbt, eps, mom), values_bn (same 7).
    Returns (keys_res_0, values_0, keys_res_1, values_1, ...)."""

    PER_HEAD = 15

    @staticmethod
    def forward(ctx, n, group, x, *args):
        # n < 0: -n heads AND x itself as the first output (the block's identity shortcut, layers/multihead_ct.py:170-176:
        # its cotangent comes back here and rides the data gradient's epilogue instead of a separate add over three tensors)
        ctx.passthrough = n < 0
        n = abs(n)
        P = UnionKeysValuesFn.PER_HEAD
        heads = [args[i * P:(i + 1) * P] for i in range(n)]
        x_in = x
        x = _f32c(x)
        _dev(x)
        B, Cin, N = x.shape
        Wc = torch.cat([h[0][:, :, 0] for h in heads], dim=0)               # [sum Co, Cin]
        Ct = Wc.size(0)
        y, am_w, am_x, Wt = pw_forward(Wc, x, ctx.needs_input_grad[2])      # [B, sum Co, N]
        outs, items, meta, c0 = [], [], [], 0
        for h in heads:
            for (w, b, rm, rv, nbt, eps, mom) in (h[1:8], h[8:15]):
                w, b = _f32c(w), _f32c(b)
                C = w.numel()
                o = torch.empty(B, C, N, device=x.device, dtype=torch.float32)
                items.append(dict(x=_ptr(y) + c0 * N * 4, xbs=Ct * N, C=C, w=w, b=b, rm=rm, rv=rv, nbt=nbt, eps=eps, mom=mom,
                                  relu=0, res=None, rbs=0, y=_ptr(o), ybs=0))
                outs.append(o)
                meta.append((c0, C))
                c0 += C
        assert c0 == Ct, "key_bn + values_bn must cover the projections"
        with _on(x.device):
            stats, count = _bn_group_fwd(items, B, N, x.device, group)
        saved = []
        for it, (mean, rstd) in zip(items, stats):
            saved += [it["w"], it["b"], mean, rstd]
        ctx.save_for_backward(x, y, Wc, count, *saved)
        ctx.am = (am_w, am_x, Wt)
        ctx.meta = meta
        ctx.couts = [h[0].size(0) for h in heads]
        ctx.group = group
        if ctx.passthrough:
            return (x_in,) + tuple(outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        g_skip = None
        if ctx.passthrough:
            g_skip, gouts = gouts[0], gouts[1:]
        x, y, Wc, count = ctx.saved_tensors[:4]
        saved = ctx.saved_tensors[4:]
        B, Cin, N = x.shape
        Ct = Wc.size(0)
        g_y = torch.empty_like(y)
        slots = _amax_slots(Ct, x.device)
        items = []
        for i, (c0, C) in enumerate(ctx.meta):
            w, b, mean, rstd = saved[i * 4:(i + 1) * 4]
            gy = gouts[i]
            if gy is None:
                gy = torch.zeros(B, C, N, device=x.device, dtype=torch.float32)
            gybs = _batch_stride(gy, C, N)
            if gybs is None:
                gy, gybs = _f32c(gy), 0
            items.append(dict(x=_ptr(y) + c0 * N * 4, xbs=Ct * N, C=C, w=w, b=b, mean=mean, rstd=rstd, gy=_ptr(gy), gybs=gybs,
                              gx=_ptr(g_y) + c0 * N * 4, gxbs=Ct * N, relu=0, keep=gy,
                              amax=None if slots is None else _ptr(slots) + 4 * c0))
        with _on(x.device):
            bn_grads = _bn_group_bwd(items, B, N, x.device, ctx.group, count)
        g_x, g_Wc = pw_backward(Wc, x, g_y, ctx.am[0], ctx.am[1], ctx.needs_input_grad[2], True,
                                am_g=None if slots is None else slots.view(-1, g_y.shape[1]), Wt=ctx.am[2],
                                add_gx=g_skip if ctx.needs_input_grad[2] else None)   # g_Wc [sum Co, Cin]
        grads, r0 = [None, None, g_x], 0
        for hi, Co in enumerate(ctx.couts):
            (gwk, gbk), (gwv, gbv) = bn_grads[2 * hi], bn_grads[2 * hi + 1]
            grads += [g_Wc[r0:r0 + Co].unsqueeze(-1), gwk, gbk, None, None, None, None, None, gwv, gbv, None, None, None, None, None]
            r0 += Co
        return tuple(grads)


def union_keys_values(x, convs, key_bns, values_bns, passthrough=False):
    """[(key_bn_i(y_i[:, :Ck]), values_bn_i(y_i[:, Ck:])) with y_i = conv_i(x)] for the heads of a union block through
    UnionKeysValuesFn; the caller checked union_keys_values_eligible.  passthrough: -> (x', that list) with x' = x as an output
    of the same node — the block's identity shortcut takes x' so that its cotangent is summed inside the data gradient."""
    args = []
    for conv, kb, vb in zip(convs, key_bns, values_bns):
        args.append(conv.weight)
        for bn in (kb, vb):
            args += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum]
    outs = UnionKeysValuesFn.apply(-len(convs) if passthrough else len(convs), _sync_group(key_bns[0]), x, *args)
    if passthrough:
        return outs[0], [(outs[1 + 2 * i], outs[2 + 2 * i]) for i in range(len(convs))]
    return [(outs[2 * i], outs[2 * i + 1]) for i in range(len(convs))]


def union_keys_values_eligible(x, convs, key_bns, values_bns):
    """Plain bias-free 1x1 Conv1d projections of the same input and norms the fused kernels take."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous() and len(convs) > 1):
        return False
    for conv, kb, vb in zip(convs, key_bns, values_bns):
        if not (isinstance(conv, torch.nn.Conv1d) and conv.kernel_size == (1,) and conv.stride == (1,) and conv.padding == (0,)
                and conv.dilation == (1,) and conv.groups == 1 and conv.bias is None and conv.in_channels == x.size(1)
                and conv.out_channels == kb.num_features + vb.num_features):
            return False
        probe = (x.size(0), x.size(2))
        for bn in (kb, vb):
            if not (type(bn) in _BN_TYPES and bn.training and bn.affine and bn.track_running_stats
                    and bn.momentum is not None):
                return False
            key = (probe[0], bn.num_features, probe[1])
            ok = _bn_supported.get(key)
            if ok is None:
                ok = _bn_supported[key] = bool(_lib.load().ct_bn_relu_supported(*key))
            if not ok:
                return False
    return norms_share_group(list(key_bns) + list(values_bns))


_bn_supported = {}
# nn.BatchNorm1d, and nn.SyncBatchNorm (what parallel.data_parallel / the reference's DDP recipe turns every norm into:
# train_segmentation.py:128): same parameters and buffers, statistics exchanged over its process group
_BN_TYPES = (torch.nn.BatchNorm1d, torch.nn.SyncBatchNorm)


def bn_relu_eligible(bn, x, channels=None):
    """True when `bn` (an nn.BatchNorm1d, exactly) applied to x (or, with `channels`, to a slice of that many of its
    channels) can run as ct_bn_relu_*: training mode with running statistics and a fixed momentum, affine, CUDA fp32
    [B,C,N] contiguous, and a shape the register-resident kernels take (ct_bn_relu_supported)."""
    C = x.size(1) if channels is None and x.dim() == 3 else channels
    if not (type(bn) in _BN_TYPES and bn.training and bn.affine and bn.track_running_stats
            and bn.momentum is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
            and x.is_contiguous() and x.data_ptr() % 16 == 0 and C == bn.num_features):
        return False
    key = (x.size(0), C, x.size(2))
    ok = _bn_supported.get(key)
    if ok is None:
        ok = _bn_supported[key] = bool(_lib.load().ct_bn_relu_supported(*key))
    return ok


def split_bn(x, bn_a, bn_b):
    """(bn_a(x[:, :Ca]), bn_b(x[:, Ca:])) through SplitBnFn; the caller checked bn_relu_eligible for both modules (each
    against its own slice's shape) and that x is contiguous."""
    return SplitBnFn.apply(x, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var, bn_a.num_batches_tracked,
                           bn_a.eps, bn_a.momentum,
                           bn_b.weight, bn_b.bias, bn_b.running_mean, bn_b.running_var, bn_b.num_batches_tracked,
                           bn_b.eps, bn_b.momentum, _sync_group(bn_a))


def bn_relu(x, bn, relu=True, residual=None):
    """relu?(bn(x)) [+ residual] through the fused kernels; the caller checked bn_relu_eligible(bn, x).  Updates the
    module's running statistics and num_batches_tracked exactly as nn.BatchNorm1d.forward does in training mode."""
    return BnReluFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                          bn.momentum, relu, residual, _sync_group(bn))


# ---------------------------------------------------------------------------
# functional entry points
# ---------------------------------------------------------------------------
def positions(keys, tensor_size, heads, dim):
    if keys.numel() == 0:
        _dev(keys)
        B, _, N = keys.shape
        V = 1 << dim
        lc = torch.zeros(B, heads, V, N, device=keys.device, dtype=torch.float32)
        e = _empty(keys)
        return (lc if e is None else lc + e), torch.zeros(B, heads, V, N, device=keys.device, dtype=torch.int64)
    return PositionsFn.apply(keys, sizes_of(tensor_size, dim), heads)


def _empty(*tensors):
    """Empty batches / empty clouds never reach the kernels (the ABI rejects zero sizes): their result is
    defined by the op itself — an empty cloud rasterises to the zero floor, slices to an empty tensor —
    and stays connected to the autograd graph with zero cotangents."""
    z = None
    for t in tensors:
        if t is not None and t.requires_grad:
            z = t.sum() * 0 if z is None else z + t.sum() * 0
    return z


def splat_keys(keys, features, pts_padding, tensor_size, heads, dim, reduce="max"):
    W = sizes_of(tensor_size, dim)
    if features.numel() == 0:
        _dev(keys, features)
        out = torch.zeros(features.shape[0], features.shape[1], *W, device=features.device, dtype=torch.float32)
        e = _empty(keys, features)
        return out if e is None else out + e
    return SplatKeysFn.apply(keys, features, pts_padding, W, heads, reduce)


def slice_keys(keys, grid, pts_padding, tensor_size, heads, dim):
    W = sizes_of(tensor_size, dim)
    if keys.numel() == 0 or grid.numel() == 0:
        _dev(keys, grid)
        out = torch.zeros(grid.shape[0], grid.shape[1], keys.shape[-1], device=grid.device, dtype=torch.float32)
        e = _empty(keys, grid)
        return out if e is None else out + e
    return SliceKeysFn.apply(keys, grid, pts_padding, W, heads)


def splat_lc(lc, idx, features, pts_padding, tensor_size, heads, dim, reduce="max"):
    W = sizes_of(tensor_size, dim)
    if features.numel() == 0:
        _dev(lc, features)
        out = torch.zeros(features.shape[0], features.shape[1], *W, device=features.device, dtype=torch.float32)
        e = _empty(lc, features)
        return out if e is None else out + e
    return SplatLcFn.apply(lc, idx, features, pts_padding, W, heads, reduce)


def slice_lc(lc, idx, grid, pts_padding, tensor_size, heads, dim):
    W = sizes_of(tensor_size, dim)
    if lc.numel() == 0 or grid.numel() == 0:
        _dev(lc, grid)
        out = torch.zeros(grid.shape[0], grid.shape[1], lc.shape[-1], device=grid.device, dtype=torch.float32)
        e = _empty(lc, grid)
        return out if e is None else out + e
    return SliceLcFn.apply(lc, idx, grid, pts_padding, W, heads)


def grid_occupancy_count(grid):
    """Number of elements with |z| > 1e-9 as a 0-dim int64 device tensor
    (layers/multihead_ct.py:104-105 divides it by B*C*H)."""
    _dev(grid)
    grid = _f32c(grid)
    count = torch.empty((), device=grid.device, dtype=torch.int64)
    lib = _lib.load()
    with _on(grid.device):
        _lib.check(lib.ct_grid_occupancy(_ptr(grid), grid.numel(), _ptr(count), _stream()), "ct_grid_occupancy")
    return count


_occ_ws = {}


def occupancy_scale(denominator):
    """float32(1) / float32(K) as a Python float: the scalar torch multiplies by when a float tensor is divided by the number K."""
    import numpy as np
    return float(np.float32(1.0) / np.float32(denominator))



def grid_occupancy_ratio(grid, denominator):
    """count(|z| > 1e-9) / denominator as a 0-dim float32 device tensor — the `occ` statistic of a block
    (layers/multihead_ct.py:104-105) — in ONE launch (ct_grid_occupancy_ratio): `grid_occupancy_count(z).float() / K` is a
    memset node, a kernel and two elementwise launches per head and forward.  Same arithmetic as torch's: float(count) * (1 / K)."""
    _dev(grid)
    grid = _f32c(grid)
    key = (grid.device.index, _stream(grid.device))
    ws = _occ_ws.get(key)
    if ws is None:          # zeroed once; the kernel hands its ticket back as zero; one workspace per stream (ordered launches)
        ws = _occ_ws[key] = torch.zeros(_lib.OCC_WORKSPACE_BYTES // 8, device=grid.device, dtype=torch.int64)
    out = torch.empty((), device=grid.device, dtype=torch.float32)
    inv = occupancy_scale(denominator)
    lib = _lib.load()
    with _on(grid.device):
        _lib.check(lib.ct_grid_occupancy_ratio(_ptr(grid), grid.numel(), inv, _ptr(out), _ptr(ws), _stream()),
                   "ct_grid_occupancy_ratio")
    return out


# ---------------------------------------------------------------------------
# plane-resident MHCT core: Splat -> grouped conv -> Slice in one kernel
# ---------------------------------------------------------------------------
# The blocks route their Splat -> conv -> Slice core through the plane-resident kernel where it is built (the shapes of
# ct_mhct_core_supported); CLOUDCT_FUSED_CORE=0 (or ops.FUSED_CORE = False) keeps the three-kernel chain, for A/B runs.
import os as _os
FUSED_CORE = _os.environ.get("CLOUDCT_FUSED_CORE", "1") != "0"
# the LDS-resident backward (16^2 C16 planes: ct_mhct_core_bwd_fused).  Built, parity-tested and MEASURED SLOWER than the backward
# from saved grids (114 vs 82 us at B8 H16 N4096, 74 vs 79 at N2048, tools/core_bwd_bench.py: one workgroup per plane moves
# 5 x 256 KiB through one CU) — so it is off unless CLOUDCT_FUSED_CORE_BWD=1
FUSED_CORE_BWD = _os.environ.get("CLOUDCT_FUSED_CORE_BWD", "0") == "1"
_core_supported = {}


def mhct_core_supported(B, H, C, N, W):
    key = (B, H, C, N, tuple(W))
    ok = _core_supported.get(key)
    if ok is None:
        ok = _core_supported[key] = bool(_lib.load().ct_mhct_core_supported(B, H, C, N, len(W), _lib.int_array(W)))
    return ok


_core_ws = {}


def mhct_core_workspace(device, B, H, C, N, W):
    """The forward's exchange workspace for a shape, allocated and initialised (counters zeroed) ONCE per device, STREAM and
    shape: every launch leaves the counters zeroed, and launches on one stream are ordered, so the buffer is reused — two
    blocks of one shape running side by side on two streams (the heads of a union block, user code) get a buffer each."""
    key = (device.index, _stream(device), B, H, C, N, tuple(W))
    ws = _core_ws.get(key)
    if ws is None:
        lib = _lib.load()
        Wa = _lib.int_array(W)
        n = lib.ct_mhct_core_workspace_bytes(B, H, C, N, len(W), Wa)
        ws = torch.empty(n, device=device, dtype=torch.uint8)
        with _on(device):
            _lib.check(lib.ct_mhct_core_workspace_init(_ptr(ws), n, B, H, C, N, len(W), Wa, _stream()), "ct_mhct_core_workspace_init")
        _core_ws[key] = ws
    return ws


def mhct_core_check(raise_on_fault=True):
    """Did a cluster of any ct_mhct_core_fwd launch so far give up waiting for its partners (include/cloudct.h: status word of
    the workspace)?  Reads the status of every cached workspace — one small copy each, AFTER synchronising: call it at points
    that wait for the device anyway (harness.fit's logging flush).  A workspace whose word is set is re-initialised; with
    `raise_on_fault` the first such workspace raises (the affected launch wrote NaN where its results would have gone)."""
    lib = _lib.load()
    bad = []
    for key, ws in list(_core_ws.items()):
        dev_index, _stream_id, B, H, C, N, W = key
        Wa = _lib.int_array(W)
        st = ctypes.c_int(0)
        with torch.cuda.device(dev_index):
            _lib.check(lib.ct_mhct_core_status(_ptr(ws), ws.numel(), B, H, C, N, len(W), Wa, ctypes.byref(st),
                                               _stream()), "ct_mhct_core_status")
            if st.value != 0:
                bad.append(key)
                _lib.check(lib.ct_mhct_core_workspace_init(_ptr(ws), ws.numel(), B, H, C, N, len(W), Wa,
                                                           _stream()), "ct_mhct_core_workspace_init")
    if bad and raise_on_fault:
        raise RuntimeError("ct_mhct_core_fwd: a cluster timed out waiting for its partners on %d workspace(s) %r; the affected "
                           "outputs hold NaN.  The workspaces were re-initialised." % (len(bad), bad[:2]))
    return bad


class MhctCoreFn(torch.autograd.Function):
    """out = Slice(keys, conv(Splat(keys, feat))) and the occupancy count of the rasterised grid
    (layers/multihead_ct.py:99-107).  Forward: ct_mhct_core_fwd — z and conv(z) stay in LDS; when a gradient is needed
    the kernel also writes them out once for the backward (ct_mhct_core_bwd: Slice backward, the two conv gradients,
    Splat backward with the key cotangents summed)."""

    @staticmethod
    def forward(ctx, keys, feat, pad, weight, bias, W, H):
        _dev(keys, feat, pad, weight, bias)
        keys, feat, weight = _f32c(keys), _f32c(feat), _f32c(weight)
        bias = _f32c(bias) if bias is not None else None
        dim = len(W)
        B, HC, N = feat.shape
        C = HC // H
        assert keys.shape == (B, H * dim, N) and weight.shape[0] == HC and weight.shape[1] == C
        padt, pad_code = _pad_args(pad, B, N)
        dev = feat.device
        need_grad = any(ctx.needs_input_grad[i] for i in (0, 1, 3, 4))
        lib = _lib.load()
        Wa = _lib.int_array(W)
        # one workgroup per plane recomputes z and conv(z) in the backward where all five tiles fit a CU: nothing to save
        recompute = bool(need_grad and FUSED_CORE_BWD and B * H >= 64 and lib.ct_mhct_core_bwd_fused_supported(B, H, C, N, dim, Wa))
        out = torch.empty(B, HC, N, device=dev, dtype=torch.float32)
        z = torch.empty(B, HC, *W, device=dev, dtype=torch.float32) if need_grad and not recompute else None
        y = torch.empty(B, HC, *W, device=dev, dtype=torch.float32) if need_grad and not recompute else None
        occ = torch.empty((), device=dev, dtype=torch.int64)
        ws = mhct_core_workspace(dev, B, H, C, N, W)
        nws = ws.numel()
        with _on(dev):
            _lib.check(lib.ct_mhct_core_fwd(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(weight), _ptr(bias), _ptr(out),
                                            _ptr(z), _ptr(y), _ptr(occ), _ptr(ws), nws, B, H, C, N, dim, Wa, _stream()),
                       "ct_mhct_core_fwd")
        ctx.save_for_backward(keys, feat, padt, weight, z, y, bias if recompute else None)
        ctx.meta = (W, H, C, pad_code, bias is not None)
        ctx.mark_non_differentiable(occ)
        return out, occ

    @staticmethod
    def backward(ctx, g_out, _g_occ):
        keys, feat, padt, weight, z, y, bias = ctx.saved_tensors
        W, H, C, pad_code, has_bias = ctx.meta
        dim = len(W)
        B, HC, N = feat.shape
        dev = feat.device
        g_out = _f32c(g_out)
        g_feat = torch.empty_like(feat)
        g_keys = torch.empty_like(keys)
        g_w = torch.empty_like(weight)
        g_b = torch.empty(HC, device=dev, dtype=torch.float32) if has_bias else None
        lib = _lib.load()
        Wa = _lib.int_array(W)
        if z is None:                 # LDS-resident backward: recomputes the grids from the points
            nws = lib.ct_mhct_core_bwd_fused_workspace_bytes(B, H, C, N, dim, Wa)
            ws = torch.empty(nws, device=dev, dtype=torch.uint8)
            with _on(dev):
                _lib.check(lib.ct_mhct_core_bwd_fused(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(weight), _ptr(bias), _ptr(g_out),
                                                      _ptr(g_feat), _ptr(g_keys), _ptr(g_w), _ptr(g_b), _ptr(ws), nws,
                                                      B, H, C, N, dim, Wa, _stream()), "ct_mhct_core_bwd_fused")
            return g_keys, g_feat, None, g_w, g_b, None, None
        nws = lib.ct_mhct_core_bwd_workspace_bytes(B, H, C, N, dim, Wa)
        ws = torch.empty(nws, device=dev, dtype=torch.uint8)
        with _on(dev):
            _lib.check(lib.ct_mhct_core_bwd_tk(_ptr(keys), _ptr(feat), _ptr(padt), pad_code, _ptr(weight), _ptr(z), _ptr(y),
                                               _ptr(g_out), _ptr(g_feat), _ptr(g_keys), _ptr(g_w), _ptr(g_b), _ptr(ws), nws,
                                               _ptr(raster_tickets(dev)), B, H, C, N, dim, Wa, _stream()), "ct_mhct_core_bwd_tk")
        return g_keys, g_feat, None, g_w, g_b, None, None


def mhct_core(keys, features, pts_padding, weight, bias, tensor_size, heads, dim):
    """(sliced features, occupancy count) of one MHCT core; raises unless mhct_core_supported(...)."""
    W = sizes_of(tensor_size, dim)
    return MhctCoreFn.apply(keys, features, pts_padding, weight, bias, W, heads)

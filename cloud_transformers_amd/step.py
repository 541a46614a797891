"""Allocation-free fwd+bwd pass of positions -> Splat -> Slice through the C ABI.

This is the unit `bench.py` times and `__graft_entry__.smoke()` checks: every
buffer is allocated once, the four ABI calls enqueue on torch's current stream,
so a whole step can be captured into a HIP graph and replayed.
"""
import torch

from . import _lib
from .ops import _ptr, _stream, sizes_of


class SplatSliceStep:
    """out = Slice(keys, Splat(keys, feat)); backward from the cotangent `cot`.

    forward : z = ct_splat_fwd(keys, feat);  out = ct_slice_fwd(keys, z)
    backward: g_z, g_keys_b = ct_slice_bwd(keys, z, cot)
              g_feat, g_keys_a = ct_splat_bwd(keys, feat, z, g_z)
    """

    def __init__(self, keys, feat, cot, tensor_size, heads, dim, reduce="max"):
        assert keys.is_cuda and feat.is_cuda and cot.is_cuda
        self.W = sizes_of(tensor_size, dim)
        self.H, self.dim, self.reduce = heads, dim, reduce
        self.B, HC, self.N = feat.shape
        self.C = HC // heads
        self.keys, self.feat, self.cot = keys.contiguous(), feat.contiguous(), cot.contiguous()
        dev = feat.device
        self.z = torch.empty(self.B, HC, *self.W, device=dev)
        self.out = torch.empty_like(self.feat)
        self.g_z = torch.empty_like(self.z)
        self.g_feat = torch.empty_like(self.feat)
        self.g_keys_a = torch.empty_like(self.keys)
        self.g_keys_b = torch.empty_like(self.keys)
        self.lib = _lib.load()
        self.Wa = _lib.int_array(self.W)
        self.red = _lib.REDUCE[reduce]
        nws = self.lib.ct_splat_bwd_workspace_bytes(self.B, self.H, self.C, self.N, dim, self.Wa, self.red)
        self.ws = torch.empty(nws, device=dev, dtype=torch.uint8) if nws else None
        self.nws = nws

    # the four passes, individually callable (bench.py times them one by one)
    def splat_fwd(self):
        _lib.check(self.lib.ct_splat_fwd(_ptr(self.keys), _ptr(self.feat), None, 0, _ptr(self.z),
                                         self.B, self.H, self.C, self.N, self.dim, self.Wa, self.red, _stream()),
                   "ct_splat_fwd")

    def slice_fwd(self):
        _lib.check(self.lib.ct_slice_fwd(_ptr(self.keys), _ptr(self.z), None, 0, _ptr(self.out),
                                         self.B, self.H, self.C, self.N, self.dim, self.Wa, _stream()),
                   "ct_slice_fwd")

    def slice_bwd_grid(self):
        _lib.check(self.lib.ct_slice_bwd_grid(_ptr(self.keys), None, 0, _ptr(self.cot), _ptr(self.g_z),
                                              self.B, self.H, self.C, self.N, self.dim, self.Wa, _stream()),
                   "ct_slice_bwd_grid")

    def slice_bwd_keys(self):
        _lib.check(self.lib.ct_slice_bwd_keys(_ptr(self.keys), _ptr(self.z), None, 0, _ptr(self.cot),
                                              _ptr(self.g_keys_b),
                                              self.B, self.H, self.C, self.N, self.dim, self.Wa, _stream()),
                   "ct_slice_bwd_keys")

    def slice_bwd(self):
        _lib.check(self.lib.ct_slice_bwd(_ptr(self.keys), _ptr(self.z), None, 0, _ptr(self.cot),
                                         _ptr(self.g_z), _ptr(self.g_keys_b),
                                         self.B, self.H, self.C, self.N, self.dim, self.Wa, _stream()),
                   "ct_slice_bwd")

    def splat_bwd(self):
        _lib.check(self.lib.ct_splat_bwd(_ptr(self.keys), _ptr(self.feat), None, 0, _ptr(self.z), _ptr(self.g_z),
                                         _ptr(self.g_feat), _ptr(self.g_keys_a), _ptr(self.ws), self.nws,
                                         self.B, self.H, self.C, self.N, self.dim, self.Wa, self.red, _stream()),
                   "ct_splat_bwd")

    # one entry per ABI call; each is ONE kernel launch for reduce="max" on the headline shape
    PASSES = ("splat_fwd", "slice_fwd", "slice_bwd", "splat_bwd")
    # HIP kernel behind each pass on the headline shape (name as rocprofv3 prints it)
    KERNELS = {
        "splat_fwd": "scatter_quad_kernel<2, false, false>",
        "slice_fwd": "quad_kernel<2, 0, 4, 512, false, false>",
        "slice_bwd": "quad_kernel<2, 1, 4, 512, true, false> + scatter_quad_kernel<2, true, false>",
        "splat_bwd": "quad_kernel<2, 2, 4, 1024, false, false>",
    }
    # passes that are exactly one kernel launch (slice_bwd is two: ~47 us + ~42 us on the headline shape)
    SINGLE_KERNEL = ("splat_fwd", "slice_fwd", "splat_bwd")

    def run(self):
        self.splat_fwd()
        self.slice_fwd()
        self.slice_bwd()
        self.splat_bwd()

    def g_keys(self):
        return self.g_keys_a + self.g_keys_b

    # algorithmic (compulsory) HBM bytes of the fused formulation, SURVEY.md §8(d)
    def algorithmic_bytes(self):
        P = self.B * self.N * self.H
        G = 1
        for w in self.W:
            G *= w
        kb = 4 * self.dim * P          # keys-sized
        fb = 4 * self.C * P            # feature-sized
        gb = 4 * self.C * G * self.B * self.H   # grid-sized
        per = {
            "splat_fwd": kb + fb + gb,                    # keys, feat -> z
            "slice_fwd": kb + gb + fb,                    # keys, z -> out
            "slice_bwd": kb + fb + gb + gb + kb,          # keys, cot, z -> g_z, g_keys
            "splat_bwd": kb + fb + gb + gb + fb + kb,     # keys, feat, z, g_z -> g_feat, g_keys
        }
        per["total"] = sum(per.values())                  # = 6*kb + 5*fb + 6*gb
        return per

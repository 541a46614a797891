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
    backward: g_z, g_keys_slice = ct_slice_bwd_tk(keys, z, cot)
              g_feat, g_keys = ct_splat_bwd_tk(keys, feat, z, g_z, add = g_keys_slice)
    The keys feed both ops, so their two key cotangents are summed (what autograd does in the module path); the
    second backward adds the first one's result inside its own store.
    """

    def __init__(self, keys, feat, cot, tensor_size, heads, dim, reduce="max", tickets=True, plane_sort=False):
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
        self.g_keys_buf = torch.empty_like(self.keys)        # Slice's key cotangent
        self.g_keys_out = self.g_keys_buf                    # + Splat's: added in place ...
        self.lib = _lib.load()
        self.Wa = _lib.int_array(self.W)
        self.red = _lib.REDUCE[reduce]
        nws = self.lib.ct_splat_bwd_ex_workspace_bytes(self.B, self.H, self.C, self.N, dim, self.Wa, self.red,
                                                       _lib.BWD_ACCUMULATE_KEYS)
        self.ws = torch.empty(nws, device=dev, dtype=torch.uint8) if nws else None
        self.nws = nws
        nws2 = self.lib.ct_slice_bwd_workspace_bytes(self.B, self.H, self.C, self.N, dim, self.Wa)
        self.ws2 = torch.empty(nws2, device=dev, dtype=torch.uint8) if nws2 else None
        self.nws2 = nws2
        # arrival tickets (ct_tickets_init contract: zero once, the kernels leave them zero): the sums over a plane's
        # workgroups happen inside the backward kernels — one launch per pass on the few-plane shapes too
        self.tickets = torch.zeros(_lib.TICKETS_BYTES // 4, device=dev, dtype=torch.int32) if tickets else None
        # sorted planes (ct_plane_sort): one record per key tensor, read by the backward passes (None: the kernels that use the
        # sorted form sort inside — measured faster where a plane is ONE workgroup: 190.7 vs 200.9 us per headline step,
        # profiles/r5_step_overlap.txt; the record pays where several workgroups share a plane)
        nps = self.lib.ct_plane_sort_bytes(self.B, self.H, self.N, dim, self.Wa) if plane_sort else 0
        self.sorted = torch.empty(nps, device=dev, dtype=torch.uint8) if nps else None
        self.nps = nps
        if tickets and reduce == "max" and self.lib.ct_splat_bwd_tk_segments(self.B, self.H, self.C, self.N, dim, self.Wa) > 1:
            self.g_keys_out = torch.empty_like(self.keys)    # ... or into a second tensor where point segments pay (ct_splat_bwd_tk)

    # the four passes, individually callable (bench.py times them one by one)
    def plane_sort(self):
        if self.sorted is not None:
            _lib.check(self.lib.ct_plane_sort(_ptr(self.keys), _ptr(self.sorted), self.nps, self.B, self.H, self.N, self.dim,
                                              self.Wa, _stream()), "ct_plane_sort")

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
                                              _ptr(self.g_keys_buf),
                                              self.B, self.H, self.C, self.N, self.dim, self.Wa, _stream()),
                   "ct_slice_bwd_keys")

    def slice_bwd(self):
        _lib.check(self.lib.ct_slice_bwd_ps(_ptr(self.keys), _ptr(self.z), None, 0, _ptr(self.cot),
                                            _ptr(self.g_z), _ptr(self.g_keys_buf), _ptr(self.ws2), self.nws2, _ptr(self.tickets),
                                            _ptr(self.sorted), self.B, self.H, self.C, self.N, self.dim, self.Wa, _stream()),
                   "ct_slice_bwd_ps")

    def splat_bwd(self):
        """g_keys_out = g_keys_buf (Slice's key cotangent: call after slice_bwd) + Splat's"""
        _lib.check(self.lib.ct_splat_bwd_tk(_ptr(self.keys), _ptr(self.feat), None, 0, _ptr(self.z), _ptr(self.g_z),
                                            _ptr(self.g_feat), _ptr(self.g_keys_buf), _ptr(self.g_keys_out), _ptr(self.ws), self.nws,
                                            _ptr(self.tickets), self.B, self.H, self.C, self.N, self.dim, self.Wa, self.red,
                                            _stream()),
                   "ct_splat_bwd_tk")

    # one entry per ABI call
    PASSES = ("splat_fwd", "slice_fwd", "slice_bwd", "splat_bwd")

    # kernel-family tag (ct_debug_last_launch) -> kernel name as rocprofv3 prints it (substring)
    KERNEL_OF = {
        "scatter_quad_max": "scatter_quad_kernel<2, false, false, 32>",
        "scatter_add_fx_reg": "scatter_add_fx_reg_kernel",
        "scatter_add_fused": "slice_bwd_fused_kernel<false, 32, 2, false>",
        "scatter_add_sorted": "slice_bwd_sorted_kernel<false, 32, false, false",
        "splat_sum_bwd_hot": "splat_sum_bwd_kernel",
        "gather_ci": "gather_ci_kernel",
        "gather_quad": "quad_kernel<2, 0,",
        "slice_bwd_fused": "slice_bwd_fused_kernel",
        "slice_bwd_sorted": "slice_bwd_sorted_kernel<false, 32, false, true",
        "slice_bwd_presorted": "slice_bwd_sorted_kernel<false, 32, true, true",
        "splat_max_bwd_hot": "splat_max_bwd_hot_kernel",
        "splat_max_bwd_whole_head": "quad_kernel<2, 2, 4, 1024,",
    }

    # the four kernels of the headline workload (B8 N4096 H64 C16 32^2 max) by their FULL instantiated names, as rocprofv3 prints
    # them — what bench.py looks up in the committed PMC record (profiles/traffic_latest.json); tests/test_headline_gpu.py holds the
    # dispatch to the kernel families behind them
    HEADLINE_KERNELS = {
        "splat_fwd": "scatter_quad_kernel<2, false, false, 32>",
        "slice_fwd": "gather_ci_kernel<false, 32>",
        "slice_bwd": "slice_bwd_sorted_kernel<false, 32, false, true, 4096, 16>",
        "splat_bwd": "splat_max_bwd_hot_kernel<false, 32, 2, 512>",
    }

    def launch_tags(self):
        """{pass: kernel-family tags of the launches behind it} for this shape (runs every pass once)."""
        tags = {}
        for name in self.PASSES:
            getattr(self, name)()
            tags[name] = self.lib.ct_debug_last_launch().decode()
        return tags

    def run(self):
        self.plane_sort()
        self.splat_fwd()
        self.slice_fwd()
        self.slice_bwd()
        self.splat_bwd()

    def g_keys(self):
        """d(out . cot)/d(keys) after run(): Slice's and Splat's key cotangents, already summed"""
        return self.g_keys_out

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

"""CPU stand-ins for the four statistics-exchange norm kernels of csrc/ct_bnorm.hip (ct_bn_stats_fwd, ct_bn_apply_fwd,
ct_bn_reduce_bwd, ct_bn_apply_bwd) and their group launches (ct_bn_group_stats_fwd ...), in float64 on the same raw pointers
and strides.

TEST INFRASTRUCTURE ONLY (tests/test_syncbn_gloo.py): they let the product's HOST logic — buffer layout, offsets,
the one-all_gather / one-all_reduce exchange of cloud_transformers_amd/ops._bn_group_fwd / _bn_group_bwd — run on
gloo at world size 2 without a GPU.  The kernels themselves are checked on the GPU (tests/test_syncbn_gpu.py,
tests/test_bnorm_gpu.py).  Semantics follow nn.SyncBatchNorm (the reference converts every norm with
SyncBatchNorm.convert_sync_batchnorm: train_segmentation.py:128)."""
import ctypes

import numpy as np


def _vec(ptr, n):
    return np.ctypeslib.as_array(ctypes.cast(int(ptr), ctypes.POINTER(ctypes.c_float)), shape=(int(n),))


def _bcn(ptr, bs, B, C, N):
    """[B,C,N] float32 view at `ptr` with batch stride bs floats (0 = dense)"""
    bs = int(bs) or C * N
    flat = _vec(ptr, (B - 1) * bs + C * N)
    return np.lib.stride_tricks.as_strided(flat, shape=(B, C, N), strides=(bs * 4, N * 4, 4))


class FakeLib:
    """the subset of libcloudct's ABI the norm groups call when a process group is present"""

    def ct_bn_stats_fwd(self, x, xbs, mean, m2, count, B, C, N, stream):
        xv = _bcn(x, xbs, B, C, N).astype(np.float64)
        mu = xv.mean(axis=(0, 2))
        _vec(mean, C)[:] = mu
        _vec(m2, C)[:] = ((xv - mu[None, :, None]) ** 2).sum(axis=(0, 2))
        if count:
            _vec(count, 1)[0] = B * N
        return 0

    def _merge(self, g_mean, g_m2, g_count, world, stride, C):
        cnt = np.array([_vec(g_count + 4 * r * stride, 1)[0] for r in range(world)], dtype=np.float64)
        means = np.stack([_vec(g_mean + 4 * r * stride, C).astype(np.float64) for r in range(world)])
        m2s = np.stack([_vec(g_m2 + 4 * r * stride, C).astype(np.float64) for r in range(world)])
        total = cnt.sum()
        mu = (means * cnt[:, None]).sum(0) / total
        m2 = (m2s + cnt[:, None] * (means - mu) ** 2).sum(0)
        return mu, m2 / total, total

    def ct_bn_apply_fwd(self, x, xbs, w, b, g_mean, g_m2, g_count, world, stride, rm, rv, nbt, res, rbs, y, ybs,
                        save_mean, save_rstd, count_total, B, C, N, eps, mom, relu, stream):
        mu, var, total = self._merge(g_mean, g_m2, g_count, world, stride, C)
        rs = 1.0 / np.sqrt(var + eps)
        xv = _bcn(x, xbs, B, C, N).astype(np.float64)
        out = (xv - mu[None, :, None]) * (_vec(w, C) * rs)[None, :, None] + _vec(b, C)[None, :, None]
        if relu:
            out = np.maximum(out, 0.0)
        if res:
            out = out + _bcn(res, rbs, B, C, N)
        _bcn(y, ybs, B, C, N)[:] = out
        _vec(save_mean, C)[:] = mu
        _vec(save_rstd, C)[:] = rs
        if count_total:
            _vec(count_total, 1)[0] = total
        if rm:
            _vec(rm, C)[:] = (1 - mom) * _vec(rm, C) + mom * mu
            _vec(rv, C)[:] = (1 - mom) * _vec(rv, C) + mom * var * total / (total - 1)
        if nbt:
            np.ctypeslib.as_array(ctypes.cast(int(nbt), ctypes.POINTER(ctypes.c_longlong)), shape=(1,))[0] += 1
        return 0

    def _masked(self, x, xbs, w, b, mean, rstd, gy, gybs, B, C, N, relu):
        xv = _bcn(x, xbs, B, C, N).astype(np.float64)
        g = _bcn(gy, gybs, B, C, N).astype(np.float64).copy()
        mu, rs = _vec(mean, C).astype(np.float64), _vec(rstd, C).astype(np.float64)
        xh = (xv - mu[None, :, None]) * rs[None, :, None]
        if relu:
            pre = (xv - mu[None, :, None]) * (_vec(w, C) * rs)[None, :, None] + _vec(b, C)[None, :, None]
            g[~(pre > 0)] = 0.0
        return g, xh, rs

    def ct_bn_reduce_bwd(self, x, xbs, w, b, mean, rstd, gy, gybs, sum_g, sum_gx, B, C, N, relu, stream):
        g, xh, _ = self._masked(x, xbs, w, b, mean, rstd, gy, gybs, B, C, N, relu)
        _vec(sum_g, C)[:] = g.sum(axis=(0, 2))
        _vec(sum_gx, C)[:] = (g * xh).sum(axis=(0, 2))
        return 0

    def ct_bn_apply_bwd(self, x, xbs, w, b, mean, rstd, gy, gybs, sum_g, sum_gx, count, gx, gxbs, B, C, N, relu, stream):
        g, xh, rs = self._masked(x, xbs, w, b, mean, rstd, gy, gybs, B, C, N, relu)
        M = float(_vec(count, 1)[0])
        m0 = _vec(sum_g, C).astype(np.float64) / M
        m1 = _vec(sum_gx, C).astype(np.float64) / M
        _bcn(gx, gxbs, B, C, N)[:] = (_vec(w, C) * rs)[None, :, None] * (g - m0[None, :, None] - xh * m1[None, :, None])
        return 0

    # the *_amax entry points: the same passes plus the per-channel max |.| of what they wrote (cloudct.h)
    def ct_bn_apply_fwd_amax(self, x, xbs, w, b, g_mean, g_m2, g_count, world, stride, rm, rv, nbt, res, rbs, y, ybs,
                             save_mean, save_rstd, count_total, amax_out, B, C, N, eps, mom, relu, stream):
        rc = self.ct_bn_apply_fwd(x, xbs, w, b, g_mean, g_m2, g_count, world, stride, rm, rv, nbt, res, rbs, y, ybs,
                                  save_mean, save_rstd, count_total, B, C, N, eps, mom, relu, stream)
        if amax_out:
            _vec(amax_out, C)[:] = np.abs(_bcn(y, ybs, B, C, N)).max(axis=(0, 2))
        return rc

    def ct_bn_apply_bwd_amax(self, x, xbs, w, b, mean, rstd, gy, gybs, sum_g, sum_gx, count, gx, gxbs, amax_out, B, C, N, relu,
                             stream):
        rc = self.ct_bn_apply_bwd(x, xbs, w, b, mean, rstd, gy, gybs, sum_g, sum_gx, count, gx, gxbs, B, C, N, relu, stream)
        if amax_out:
            _vec(amax_out, C)[:] = np.abs(_bcn(gx, gxbs, B, C, N)).max(axis=(0, 2))
        return rc

    # the group launches (ct_bn_group_stats_fwd / _apply_fwd / _reduce_bwd / _apply_bwd, cloudct.h): every phase of all the
    # items of a group at once, on the group's buffers [mean: Ct | m2: Ct | count] resp. [sum g': Ct | sum g' xhat: Ct]
    @staticmethod
    def _items(items, n, cls):
        return ctypes.cast(int(items), ctypes.POINTER(cls * n)).contents

    def ct_bn_group_stats_fwd(self, items, n, B, N, local, stream):
        from cloud_transformers_amd._lib import BnFwdItem
        arr = self._items(items, n, BnFwdItem)
        Ct = sum(it.C for it in arr)
        c0 = 0
        for i, it in enumerate(arr):
            self.ct_bn_stats_fwd(it.x, it.x_batch_stride, local + 4 * c0, local + 4 * (Ct + c0), local + 4 * 2 * Ct if i == 0 else 0,
                                 B, it.C, N, stream)
            c0 += it.C
        return 0

    def ct_bn_group_apply_fwd(self, items, n, B, N, gathered, world, count_total, stream):
        from cloud_transformers_amd._lib import BnFwdItem
        arr = self._items(items, n, BnFwdItem)
        Ct = sum(it.C for it in arr)
        stride, c0 = 2 * Ct + 1, 0
        for i, it in enumerate(arr):
            self.ct_bn_apply_fwd_amax(it.x, it.x_batch_stride, it.weight, it.bias, gathered + 4 * c0, gathered + 4 * (Ct + c0),
                                      gathered + 4 * 2 * Ct, world, stride, it.running_mean, it.running_var, it.num_batches_tracked,
                                      it.residual, it.residual_batch_stride, it.y, it.y_batch_stride, it.save_mean, it.save_rstd,
                                      count_total if i == 0 else 0, it.amax_out, B, it.C, N, it.eps, it.momentum, it.relu, stream)
            c0 += it.C
        return 0

    def ct_bn_group_reduce_bwd(self, items, n, B, N, sums, stream):
        from cloud_transformers_amd._lib import BnBwdItem
        arr = self._items(items, n, BnBwdItem)
        Ct = sum(it.C for it in arr)
        c0 = 0
        for it in arr:
            self.ct_bn_reduce_bwd(it.x, it.x_batch_stride, it.weight, it.bias, it.save_mean, it.save_rstd, it.gy, it.gy_batch_stride,
                                  sums + 4 * c0, sums + 4 * (Ct + c0), B, it.C, N, it.relu, stream)
            c0 += it.C
        return 0

    def ct_bn_group_reduce_bwd_copy(self, items, n, B, N, sums, sums_copy, stream):
        # (the same sums twice: one buffer for the in-place collective, one stays this rank's parameter gradients)
        rc = self.ct_bn_group_reduce_bwd(items, n, B, N, sums, stream)
        rc2 = self.ct_bn_group_reduce_bwd(items, n, B, N, sums_copy, stream)
        return rc or rc2

    def ct_bn_group_apply_bwd(self, items, n, B, N, sums, count, stream):
        from cloud_transformers_amd._lib import BnBwdItem
        arr = self._items(items, n, BnBwdItem)
        Ct = sum(it.C for it in arr)
        c0 = 0
        for it in arr:
            self.ct_bn_apply_bwd_amax(it.x, it.x_batch_stride, it.weight, it.bias, it.save_mean, it.save_rstd, it.gy, it.gy_batch_stride,
                                      sums + 4 * c0, sums + 4 * (Ct + c0), count, it.gx, it.gx_batch_stride, it.amax_out, B, it.C, N,
                                      it.relu, stream)
            c0 += it.C
        return 0

    def ct_strerror(self, status):
        return b"fake"

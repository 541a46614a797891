"""Barycentre voxel-grid subsampling (KPConv-style) on the host: numpy front end of ct_grid_subsample
(include/cloudct_host.h, csrc/host/grid_subsampling.cpp).  Mirrors the reference's call surface:
`datasets/s3dis_closer.py:13-31` `grid_subsampling(points, features, labels, sampleDl, verbose)` and the extension function
behind it, `cpp_wrappers.cpp_subsampling.grid_subsampling.compute(points, features=, classes=, sampleDl=, verbose=)`
(cpp_wrappers/cpp_subsampling/wrapper.cpp)."""
import ctypes

import numpy as np

from .. import _lib


def compute(points, features=None, classes=None, sampleDl=0.1, verbose=0):
    """points f32[N,3] (+ features f32[N,F], classes i32[N] or [N,L]) -> subsampled points (, features) (, classes):
    the tuple layout of the reference's extension — one array when only points are given, else a tuple in this order."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("points must be [N, 3]")
    N = pts.shape[0]
    feats = cls = None
    fdim = ldim = 0
    if features is not None:
        feats = np.ascontiguousarray(features, dtype=np.float32)
        if feats.ndim != 2 or feats.shape[0] != N:
            raise ValueError("features must be [N, F]")
        fdim = feats.shape[1]
    cls_1d = False
    if classes is not None:
        cls = np.ascontiguousarray(classes, dtype=np.int32)
        cls_1d = cls.ndim == 1
        if cls_1d:
            cls = cls[:, None]
        if cls.ndim != 2 or cls.shape[0] != N:
            raise ValueError("classes must be [N] or [N, L]")
        cls = np.ascontiguousarray(cls)
        ldim = cls.shape[1]
    out_p = np.empty((N, 3), np.float32)
    out_f = np.empty((N, fdim), np.float32) if fdim else None
    out_c = np.empty((N, ldim), np.int32) if ldim else None
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    ptr = lambda a, t: None if a is None else a.ctypes.data_as(t)      # noqa: E731
    M = _lib.load_host().ct_grid_subsample(ptr(pts, f32p), ptr(feats, f32p), ptr(cls, i32p), N, fdim, ldim, float(sampleDl),
                                           ptr(out_p, f32p), ptr(out_f, f32p), ptr(out_c, i32p))
    if M < 0:
        raise ValueError("ct_grid_subsample: bad argument (sampleDl must be > 0)")
    res = [out_p[:M].copy()]
    if fdim:
        res.append(out_f[:M].copy())
    if ldim:
        c = out_c[:M].copy()
        res.append(c[:, 0] if cls_1d else c)
    return res[0] if len(res) == 1 else tuple(res)


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1, verbose=0):
    """datasets/s3dis_closer.py:13-31"""
    return compute(points, features=features, classes=labels, sampleDl=sampleDl, verbose=verbose)

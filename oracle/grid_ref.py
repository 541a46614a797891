"""ctypes access to oracle/_ref/libgrid_subsampling_ref.so — the REFERENCE's grid_subsampling.cpp compiled from
/root/reference (oracle/Makefile target `ref`; build container only) — TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libgrid_subsampling_ref.so")


def available():
    if os.path.exists(_SO):
        return True
    if os.path.isdir("/root/reference/cpp_wrappers"):
        subprocess.run(["make", "-C", _HERE, "-s", "ref"], check=True)
        return os.path.exists(_SO)
    return False


def compute(points, features=None, classes=None, dl=0.1):
    lib = ctypes.CDLL(_SO)
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    lib.ref_grid_subsample.restype = ctypes.c_int64
    lib.ref_grid_subsample.argtypes = [f32p, f32p, i32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, f32p, f32p, i32p]
    pts = np.ascontiguousarray(points, np.float32)
    N = pts.shape[0]
    feats = None if features is None else np.ascontiguousarray(features, np.float32)
    cls = None if classes is None else np.ascontiguousarray(classes, np.int32).reshape(N, -1)
    fdim = 0 if feats is None else feats.shape[1]
    ldim = 0 if cls is None else cls.shape[1]
    op = np.empty((N, 3), np.float32)
    of = np.empty((N, max(fdim, 1)), np.float32)
    oc = np.empty((N, max(ldim, 1)), np.int32)
    ptr = lambda a, t: None if a is None else a.ctypes.data_as(t)      # noqa: E731
    M = lib.ref_grid_subsample(ptr(pts, f32p), ptr(feats, f32p), ptr(cls, i32p), N, fdim, ldim, float(dl), ptr(op, f32p), ptr(of, f32p),
                               ptr(oc, i32p))
    return op[:M].copy(), (of[:M, :fdim].copy() if fdim else None), (oc[:M, :ldim].copy() if ldim else None)

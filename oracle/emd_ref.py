"""ctypes access to oracle/emd_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libemd_ref.so")


def _lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "emd_ref.c")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    lib = ctypes.CDLL(_SO)
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    lib.emd_ref_forward.restype = ctypes.c_int
    lib.emd_ref_forward.argtypes = [f32p, f32p, f32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]
    lib.emd_ref_last_getmax_ties.restype = ctypes.c_longlong
    lib.emd_ref_last_getmax_ties.argtypes = []
    lib.emd_ref_set_tie_lowest.restype = None
    lib.emd_ref_set_tie_lowest.argtypes = [ctypes.c_int]
    lib.emd_ref_backward.restype = None
    lib.emd_ref_backward.argtypes = [f32p, f32p, f32p, i32p, f32p, ctypes.c_int, ctypes.c_int]
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def forward(xyz1, xyz2, eps, iters):
    """numpy f32 [B,n,3] x2 -> (status, dist f32[B,n], assignment i32[B,n])"""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    B, n, _ = xyz1.shape
    dist = np.zeros((B, n), dtype=np.float32)
    ass = np.full((B, n), -1, dtype=np.int32)
    st = _lib().emd_ref_forward(_p(xyz1, ctypes.c_float), _p(xyz2, ctypes.c_float), _p(dist, ctypes.c_float),
                                _p(ass, ctypes.c_int), B, n, float(eps), int(iters))
    return st, dist, ass


def last_getmax_ties():
    """GetMax window ties seen by the last forward() (see emd_ref.c): 0 = the reference is deterministic on those inputs."""
    return int(_lib().emd_ref_last_getmax_ties())


def set_tie_lowest(flag):
    """experiment: GetMax window ties to the lowest bidder index (default: the highest) — tools/dev/emd_reference_soak.py"""
    _lib().emd_ref_set_tie_lowest(int(bool(flag)))


def backward(xyz1, xyz2, grad_dist, assignment):
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    g = np.ascontiguousarray(grad_dist, dtype=np.float32)
    a = np.ascontiguousarray(assignment, dtype=np.int32)
    B, n, _ = xyz1.shape
    out = np.zeros_like(xyz1)
    _lib().emd_ref_backward(_p(xyz1, ctypes.c_float), _p(xyz2, ctypes.c_float), _p(g, ctypes.c_float),
                            _p(a, ctypes.c_int), _p(out, ctypes.c_float), B, n)
    return out

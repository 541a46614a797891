"""ctypes access to oracle/emd_sim.c — the thread-decomposition simulation of the reference's auction EMD
(TEST INFRASTRUCTURE ONLY; see the header of emd_sim.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libemd_sim.so")

CONTRACTIONS = {"none": 0, "fma_z_fma_y": 1, "fma_z_only": 2, "fma_z_fma_x": 3}
GETMAX = {"lowest": 0, "highest": 1, "random": 2}
LIST = {"ascending": 0, "descending": 1, "random": 2}


def _lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "emd_sim.c")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    lib = ctypes.CDLL(_SO)
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    lib.emd_sim_forward.restype = ctypes.c_int
    lib.emd_sim_forward.argtypes = [f32p, f32p, f32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_longlong)]
    return lib


def forward(xyz1, xyz2, eps, iters, contraction="fma_z_fma_y", getmax="highest", list_order="ascending", seed=1):
    """numpy f32 [B,n,3] x2 -> (status, dist f32[B,n], assignment i32[B,n], {"bid_value_ties", "getmax_multi"})"""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    B, n, _ = xyz1.shape
    dist = np.zeros((B, n), dtype=np.float32)
    ass = np.full((B, n), -1, dtype=np.int32)
    stats = (ctypes.c_longlong * 2)()
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))      # noqa: E731
    st = _lib().emd_sim_forward(p(xyz1, ctypes.c_float), p(xyz2, ctypes.c_float), p(dist, ctypes.c_float), p(ass, ctypes.c_int),
                                B, n, float(eps), int(iters), CONTRACTIONS[contraction], GETMAX[getmax], LIST[list_order],
                                int(seed), stats)
    return st, dist, ass, {"bid_value_ties": int(stats[0]), "getmax_multi": int(stats[1])}

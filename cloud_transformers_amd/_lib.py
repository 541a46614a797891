"""ctypes binding of libcloudct.so (the C ABI declared in include/cloudct.h).

The shared library is built in-tree by `build()` (hipcc, gfx950 only) and loaded
lazily.  There is NO CPU fallback: if the library is missing or a tensor is not
on a HIP device the ops raise.
"""
import ctypes
import os
import subprocess
import threading

# torch must be imported BEFORE libcloudct.so is loaded: PyTorch-ROCm bundles its
# own libamdhip64; if ours pulled /opt/rocm's copy in first, the process would mix
# two HIP runtimes and every launch on a torch stream would fail.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("CLOUDCT_LIB") or os.path.join(LIB_DIR, "libcloudct.so")   # override: A/B builds
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

HIP_SOURCES = ["ct_raster.hip", "ct_mhct.hip", "ct_lattice.hip", "ct_gconv.hip", "ct_chamfer.hip", "ct_emd.hip",
               "ct_adain.hip", "ct_bnorm.hip", "ct_pwgemm.hip"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
               # index/weight math must round exactly like the reference's fp32 op sequence
               "-ffp-contract=off"]

class BnFwdItem(ctypes.Structure):
    """ct_bn_fwd_item (include/cloudct.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("x_batch_stride", ctypes.c_longlong), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p), ("num_batches_tracked", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("residual_batch_stride", ctypes.c_longlong), ("y", ctypes.c_void_p),
                ("y_batch_stride", ctypes.c_longlong), ("save_mean", ctypes.c_void_p), ("save_rstd", ctypes.c_void_p),
                ("amax_out", ctypes.c_void_p), ("C", ctypes.c_int), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
                ("relu", ctypes.c_int)]


class BnBwdItem(ctypes.Structure):
    """ct_bn_bwd_item (include/cloudct.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("x_batch_stride", ctypes.c_longlong), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("save_mean", ctypes.c_void_p), ("save_rstd", ctypes.c_void_p), ("gy", ctypes.c_void_p),
                ("gy_batch_stride", ctypes.c_longlong), ("gx", ctypes.c_void_p), ("gx_batch_stride", ctypes.c_longlong),
                ("g_weight", ctypes.c_void_p), ("g_bias", ctypes.c_void_p), ("amax_out", ctypes.c_void_p), ("C", ctypes.c_int),
                ("relu", ctypes.c_int)]


class AdainFwdItem(ctypes.Structure):
    """ct_adain_fwd_item (include/cloudct.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("x_batch_stride", ctypes.c_longlong), ("gamma_beta", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("residual_batch_stride", ctypes.c_longlong), ("y", ctypes.c_void_p),
                ("y_batch_stride", ctypes.c_longlong), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("amax_out", ctypes.c_void_p), ("amax_batch_stride", ctypes.c_longlong), ("C", ctypes.c_int), ("eps", ctypes.c_float),
                ("relu", ctypes.c_int), ("gamma_beta_batch_stride", ctypes.c_longlong)]


class AdainBwdItem(ctypes.Structure):
    """ct_adain_bwd_item (include/cloudct.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("x_batch_stride", ctypes.c_longlong), ("gamma_beta", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p), ("gy", ctypes.c_void_p), ("gy_batch_stride", ctypes.c_longlong),
                ("gx", ctypes.c_void_p), ("gx_batch_stride", ctypes.c_longlong), ("g_gamma_beta", ctypes.c_void_p),
                ("amax_out", ctypes.c_void_p), ("amax_batch_stride", ctypes.c_longlong), ("C", ctypes.c_int), ("relu", ctypes.c_int),
                ("gamma_beta_batch_stride", ctypes.c_longlong)]


ABI_VERSION = 2      # CT_ABI_VERSION of include/cloudct.h
BN_GROUP_MAX = 8
CT_OK = 0
REDUCE = {"max": 0, "sum": 1}
PAD_NONE, PAD_F32, PAD_I32 = 0, 1, 2
BWD_ACCUMULATE_KEYS = 1
TICKETS_BYTES = 65536
OCC_WORKSPACE_BYTES = 4096
DEBUG_NO_HOT = 1
DEBUG_FORCE_HOT = 2
DEBUG_NO_BAND = 4
DEBUG_FORCE_BAND = 8
DEBUG_NO_SORTED = 16
DEBUG_FORCE_SORTED = 32
DEBUG_FORCE_SORTED_SEG = 64

_lock = threading.Lock()
_lib = None


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "cloudct.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_and_link(force, verbose):
    """One object per source (lib/obj/*.o, recompiled only when the source or a header is newer; up to 4 hipcc at a
    time), then one link into a per-process temporary that is renamed into place."""
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "cloudct.h")]
    h_time = max(os.path.getmtime(h) for h in headers)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for name in HIP_SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(obj_dir, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), h_time):
            jobs.append([_hipcc()] + flags + ["-c", "-I", INCLUDE, src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(run, jobs))
    tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())
    try:
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp])
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build(force=False, verbose=False):
    """Compile every HIP source into lib/libcloudct.so (cross-compiles without a GPU).

    Safe when several processes call it at once (the ranks `launch.spawn_ranks` / torchrun start on a fresh checkout):
    the compile runs under an exclusive file lock into a per-process temporary name, staleness is re-checked under the
    lock (a rank that waited finds the library its sibling built), and the finished file is renamed into place."""
    if os.environ.get("CLOUDCT_LIB"):
        # an experiment library named by the caller: use it as it is, never relink the product objects over it
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("CLOUDCT_LIB=%s does not exist" % LIB_PATH)
        return LIB_PATH
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB_PATH
            _compile_and_link(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


HOST_SOURCES = [os.path.join("host", "grid_subsampling.cpp")]
HOST_LIB_PATH = os.path.join(LIB_DIR, "libcloudct_host.so")
_host = None


def build_host(force=False):
    """Compile the CPU-side data preparation (csrc/host/*.cpp, include/cloudct_host.h) with g++ into lib/libcloudct_host.so."""
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(INCLUDE, "cloudct_host.h")]
    if not force and os.path.exists(HOST_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB_PATH) for d in deps):
        return HOST_LIB_PATH
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build_host.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            tmp = "%s.%d.tmp" % (HOST_LIB_PATH, os.getpid())
            subprocess.run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", INCLUDE]
                           + srcs + ["-o", tmp], check=True)
            os.replace(tmp, HOST_LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return HOST_LIB_PATH


def load_host():
    """libcloudct_host.so (built on first use: g++ only, no GPU toolchain needed)."""
    global _host
    if _host is None:
        with _lock:
            if _host is None:
                lib = ctypes.CDLL(build_host())
                f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
                lib.ct_grid_subsample.restype = ctypes.c_int64
                lib.ct_grid_subsample.argtypes = [f32p, f32p, i32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                  f32p, f32p, i32p]
                _host = lib
    return _host


_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_ll = ctypes.c_longlong
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/cloudct.h one to one
SIGNATURES = {
    "ct_abi_version": (_i, []),
    "ct_strerror": (ctypes.c_char_p, [_i]),
    "ct_positions_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ip, _vp]),
    "ct_positions_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ip, _vp]),
    "ct_splat_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _ip, _i, _vp]),
    "ct_splat_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip, _i]),
    "ct_splat_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _i, _vp]),
    "ct_splat_bwd_ex_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip, _i, _i]),
    "ct_splat_bwd_ex": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _i, _i, _vp]),
    "ct_debug_set_flags": (None, [ctypes.c_uint]),
    "ct_debug_set_nseg": (None, [_i]),
    "ct_debug_last_launch": (ctypes.c_char_p, []),
    "ct_slice_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_slice_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_slice_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip]),
    "ct_slice_bwd_ws": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_tickets_init": (_i, [_vp, _vp]),
    "ct_splat_bwd_tk_segments": (_i, [_i, _i, _i, _i, _i, _ip]),
    "ct_slice_bwd_tk": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_plane_sort_bytes": (_sz, [_i, _i, _i, _i, _ip]),
    "ct_plane_sort": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _ip, _vp]),
    "ct_slice_bwd_ps": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_splat_bwd_tk": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _ip, _i, _vp]),
    "ct_slice_bwd_grid": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_slice_bwd_keys": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_splat_lc_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _ip, _i, _vp]),
    "ct_splat_lc_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _i, _vp]),
    "ct_slice_lc_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_slice_lc_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_grid_occupancy": (_i, [_vp, ctypes.c_int64, _vp, _vp]),
    "ct_grid_occupancy_ratio": (_i, [_vp, ctypes.c_int64, _f, _vp, _vp, _vp]),
    "ct_lattice_fwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "ct_lattice_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    "ct_lattice_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "ct_lattice_bwd": (_i, [_vp] * 15 + [_vp, _sz, _i, _i, _i, _i, _vp]),
    "ct_lattice_so3_fwd": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _vp]),
    "ct_lattice_so3_bwd": (_i, [_vp, _vp, _vp, _f] + [_vp] * 14 + [_vp, _sz, _i, _i, _i, _i, _vp]),
    "ct_so3_exp_fwd": (_i, [_vp, _vp, _i, _f, _vp]),
    "ct_so3_exp_bwd": (_i, [_vp, _vp, _vp, _i, _f, _vp]),
    "ct_adain_fwd": (_i, [_vp, _ll, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "ct_adain_bwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i, _vp]),
    "ct_adain_fwd_amax": (_i, [_vp, _ll, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _f, _i, _vp]),
    "ct_adain_bwd_amax": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _vp]),
    "ct_bn_relu_supported": (_i, [_i, _i, _i]),
    "ct_bn_relu_fwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "ct_bn_relu_bwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ct_adain_group_fwd": (_i, [_vp, _i, _i, _i, _vp]),
    "ct_adain_group_bwd": (_i, [_vp, _i, _i, _i, _vp]),
    "ct_bn_group_fwd": (_i, [_vp, _i, _i, _i, _vp]),
    "ct_bn_group_bwd": (_i, [_vp, _i, _i, _i, _vp]),
    "ct_bn_group_stats_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ct_bn_group_apply_fwd": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "ct_bn_group_reduce_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ct_bn_group_reduce_bwd_copy": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "ct_bn_group_apply_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "ct_bn_relu_fwd_amax": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "ct_bn_relu_bwd_amax": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ct_bn_stats_fwd": (_i, [_vp, _ll, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ct_bn_apply_fwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp,
                             _i, _i, _i, _f, _f, _i, _vp]),
    "ct_bn_reduce_bwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ct_bn_apply_bwd": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp]),
    "ct_bn_apply_fwd_amax": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp,
                                  _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "ct_bn_apply_bwd_amax": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _i, _i, _i, _i, _vp]),
    "ct_gconv_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_gconv_bwd_data": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_gconv_supported": (_i, [_i, _i, _i, _i, _i, _ip]),
    "ct_gconv_bwd_weight_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip]),
    "ct_gconv_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_mhct_core_supported": (_i, [_i, _i, _i, _i, _i, _ip]),
    "ct_mhct_core_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip]),
    "ct_mhct_core_workspace_init": (_i, [_vp, _sz, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_mhct_core_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_mhct_core_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip]),
    "ct_mhct_core_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_mhct_core_bwd_tk": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_mhct_core_bwd_fused_supported": (_i, [_i, _i, _i, _i, _i, _ip]),
    "ct_mhct_core_bwd_fused_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _ip]),
    "ct_mhct_core_bwd_fused": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _ip, _vp]),
    "ct_debug_set_core": (None, [ctypes.c_uint]),
    "ct_debug_set_gconv": (None, [ctypes.c_uint]),
    "ct_debug_set_emd": (None, [ctypes.c_uint]),
    "ct_mhct_core_status": (_i, [_vp, _sz, _i, _i, _i, _i, _i, _ip, ctypes.POINTER(ctypes.c_int), _vp]),
    "ct_chamfer_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ct_chamfer_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ct_emd_workspace_bytes": (_sz, [_i, _i]),
    "ct_emd_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _f, _i, _vp]),
    "ct_emd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ct_amax_f32": (_i, [_vp, _ll, _vp, _vp]),
    "ct_amax_len": (_i, []),
    "ct_pw_prep_weight_partials": (_i, [_i, _i]),
    "ct_pw_prep_weight": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "ct_pw_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ct_pw_gemm": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _i, _i, _i, _i, _vp]),
    "ct_pw_gemm_rs": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _sz, _i, _i, _i, _i, _vp]),
    "ct_pw_gemm_rs_add": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _sz, _i, _i, _i, _i, _vp]),
    "ct_amax_rows_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ct_pw_prep_weight_rs": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
}


def load():
    """Return the loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(cloud_transformers_amd has no CPU or PyTorch fallback)")
            lib = ctypes.CDLL(LIB_PATH)
            lib.ct_abi_version.restype = ctypes.c_int
            if lib.ct_abi_version() != ABI_VERSION:
                raise RuntimeError("%s reports ABI version %d, this package binds version %d (include/cloudct.h): rebuild it — "
                                   "`python -c 'import __graft_entry__ as g; g.build()'`, or tools/dev/build_raster_exp.sh for an "
                                   "experimental library selected by CLOUDCT_LIB" % (LIB_PATH, lib.ct_abi_version(), ABI_VERSION))
            missing = [name for name in SIGNATURES if not hasattr(lib, name)]
            if missing:
                raise RuntimeError("%s lacks %d symbol(s) of include/cloudct.h (%s ...): a stale build — rebuild it"
                                   % (LIB_PATH, len(missing), ", ".join(missing[:4])))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(status, what):
    if status != CT_OK:
        msg = load().ct_strerror(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])

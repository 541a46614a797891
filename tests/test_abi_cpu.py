"""CPU-side checks of the boundary: the shared library builds, loads and exports
every symbol include/cloudct.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cloudct.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ct_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from cloud_transformers_amd import _lib
    _lib.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from cloud_transformers_amd import _lib
    syms = _declared_symbols()
    assert "ct_splat_fwd" in syms and "ct_slice_bwd" in syms and "ct_chamfer_fwd" in syms
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in cloudct.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version_and_strerror(lib):
    assert lib.ct_abi_version() == 2
    assert lib.ct_strerror(0) == b"ok"
    assert b"invalid" in lib.ct_strerror(-1)
    assert b"workspace" in lib.ct_strerror(-3)


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected before anything touches the device."""
    from cloud_transformers_amd import _lib
    W = _lib.int_array([32, 32])
    assert lib.ct_splat_fwd(None, None, None, 0, None, 1, 1, 1, 1, 2, W, 0, None) == -1
    assert lib.ct_slice_fwd(None, None, None, 0, None, 1, 1, 1, 1, 2, W, None) == -1
    assert lib.ct_positions_fwd(None, None, None, 1, 1, 1, 4, W, None) == -1      # dim = 4
    assert lib.ct_chamfer_fwd(None, None, None, None, None, None, 1, 1, 1, None) == -1
    one = _lib.int_array([1, 8])                                                   # W < 2 is rejected
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.ct_positions_fwd(p, p, p, 1, 1, 1, 2, one, None) == -1
    # workspace query is pure host arithmetic
    assert lib.ct_splat_bwd_workspace_bytes(8, 64, 8, 100, 2, W, 0) == 0            # enough (b,h) planes: one workgroup per plane
    # few planes: the launch is split into 2 channel-chunk groups, each stores a partial g_keys / g_lc (2^dim x N per plane)
    assert lib.ct_splat_bwd_workspace_bytes(8, 16, 8, 100, 2, W, 0) == 2 * (8 * 16 * 4 * 100 * 4)
    small = lib.ct_splat_bwd_workspace_bytes(2, 4, 8, 100, 2, W, 0)                  # very few planes: thinner chunks, more groups
    assert small > 0 and small % (2 * 4 * 4 * 100 * 4) == 0
    big = _lib.int_array([64, 64, 64])
    assert lib.ct_splat_bwd_workspace_bytes(2, 4, 8, 100, 3, big, 0) == 2 * 4 * 8 * 64 ** 3 * 4


def test_norm_and_rotation_layers_refuse_cpu_tensors():
    """No CPU composition behind the layers either: AdaIn1dUpd and the so3 exponential map raise off the GPU."""
    from cloud_transformers_amd.layers.utils import AdaIn1dUpd, so3_exponential_map
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AdaIn1dUpd(4, 8)(torch.zeros(2, 4, 16), torch.zeros(2, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        so3_exponential_map(torch.zeros(3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        so3_exponential_map(torch.zeros(3, 3, dtype=torch.float64))


def test_ops_refuse_cpu_tensors():
    from cloud_transformers_amd import ops
    from cloud_transformers_amd.chamfer import chamfer_with_indices
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.positions(torch.zeros(1, 2, 4), 8, 1, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        chamfer_with_indices(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))


def test_module_state_dict_names():
    """Checkpoint compatibility: every DifferentiableGridModule contributes a
    `tensor_mod` buffer of shape [1, dim, 1] (layers/cloud_transform.py:48-51)."""
    from cloud_transformers_amd.layers.cloud_transform import DifferentiablePositions, Splat, Slice
    for cls in (DifferentiablePositions, Splat, Slice):
        m = cls(tensor_size=(16, 24), heads=3, dim=2)
        sd = m.state_dict()
        assert list(sd) == ["tensor_mod"]
        assert sd["tensor_mod"].shape == (1, 2, 1) and sd["tensor_mod"].flatten().tolist() == [16.0, 24.0]
        assert m.spread_size == 4 and m.tensor_size == [16, 24]
    assert Splat(8, 2, 3).spread_size == 8


def test_block_state_dict_keys_match_reference_checkpoints():
    """Every MHCT block exposes exactly the parameter/buffer names of the reference
    (tests/golden/blocks.npz holds the reference's own state dicts)."""
    from tests.conftest import load_golden
    from cloud_transformers_amd.layers import multihead_ct as M
    g = load_golden("blocks")

    def keys(d):
        return sorted(k[3:] for k in d if k.startswith("sd/"))

    D = 32
    built = {
        "mh2d": M.MultiHead(D, 4, D, 16, 2, 4),
        "mh3d": M.MultiHead(D, 4, D, 8, 3, 2, scales=True),
        "pool": M.MultiHeadPool(D, 4, 8, 3, 2),
        "adain": M.MultiHeadAdaIn(D, 4, D, 16, 2, 4, n_latent=24),
        "union": M.MultiHeadUnion(D, [4, 4], [16, 8], [2, 3], [4, 2]),
        "union_proj": M.MultiHeadUnion(D, [4, 4], [16, 8], [2, 3], [4, 2], model_dim_out=48),
        "union_adain": M.MultiHeadUnionAdaIn(D, [4, 4], [16, 8], [2, 3], [4, 2], n_latent=24),
    }
    for case, mod in built.items():
        sd = mod.state_dict()
        assert sorted(sd) == keys(g[case]), case
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(g[case]["sd/" + k].shape), (case, k)
    # key_bn starts at zero: keys are initially the pure rigid transform of xyz
    assert float(built["mh2d"].key_bn.weight.detach().abs().max()) == 0.0
    assert float(built["adain"].scale.detach()) == 0.0


def test_reference_import_paths_resolve():
    import layers.cloud_transform as a
    import layers.multihead_ct as b
    import layers.multihead_ct_adain as c
    import layers.multihead_ct_pool as d
    import layers.utils as e
    import layers.v2v_groups as f
    import unet2d.unet_parts as g
    import chamfer_extension.dist_chamfer as h
    assert a.Splat and a.Slice and a.DifferentiablePositions
    assert b.MultiHead and b.MultiHeadUnion and c.MultiHeadAdaIn and c.MultiHeadUnionAdaIn and c.forward_style
    assert d.MultiHeadPool and e.PlaneTransformer and e.VolTransformer and e.AdaIn1dUpd
    assert f.Res3DBlock and f.Pool3DBlock and g.Res2DBlock and h.loss_chamfer and h.ChamferDist
    assert "AdaIn1dUpd" in str(type(e.AdaIn1dUpd(4, 8)))


def test_debug_flag_constants_match_the_header():
    """The CT_DEBUG_* bits of include/cloudct.h and the DEBUG_* constants the Python layer passes to ct_debug_set_flags are the
    same numbers (a test that forces a kernel family with a stale constant would silently test another one)."""
    import re
    from cloud_transformers_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cloudct.h")).read()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define CT_DEBUG_(\w+)\s+(\d+)", text)}
    assert len(header) >= 7 and len(set(header.values())) == len(header)
    for name, value in header.items():
        assert getattr(_lib, "DEBUG_" + name) == value, name
    assert _lib.ABI_VERSION == int(re.search(r"#define CT_ABI_VERSION\s+(\d+)", text).group(1))

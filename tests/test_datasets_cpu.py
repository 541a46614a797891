"""cloud_transformers_amd.data.datasets against the reference's loaders (datasets/scanobjectnn.py:87-125,
datasets/s3dis_v2.py:494-574) on synthetic files: same seeds -> same items.  The reference side runs in a child process
with a stand-in `h5py` module that serves the same arrays from .npz (h5py is not installed here; reading HDF5 is h5py's job
on both sides); build container only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def make_files(d):
    rng = np.random.default_rng(0)
    np.savez(os.path.join(d, "scan.npz"), data=rng.normal(size=(6, 256, 3)).astype(np.float32) * 0.3 + 1.0,
             label=rng.integers(0, 15, size=(6,)).astype(np.int64),
             mask=rng.integers(-1, 3, size=(6, 256)).astype(np.float32))
    for name in ("ply_data_all_0", "ply_data_all_1"):
        data = rng.random((5, 128, 9)).astype(np.float32)
        np.savez(os.path.join(d, name + ".npz"), data=data, label=rng.integers(0, 13, size=(5, 128)).astype(np.uint8))
    with open(os.path.join(d, "all_files.txt"), "w") as f:
        f.write("indoor3d_sem_seg_hdf5_data/ply_data_all_0.h5\nindoor3d_sem_seg_hdf5_data/ply_data_all_1.h5\n")
    with open(os.path.join(d, "room_filelist.txt"), "w") as f:
        f.write("\n".join(["Area_%d_office_%d" % (1 + (i % 6), i) for i in range(10)]) + "\n")


CHILD = r"""
import os, sys, types, random
import numpy as np, torch
side, d = sys.argv[1], sys.argv[2]
if side == "ref":
    h5 = types.ModuleType("h5py")
    class File(dict):
        def __init__(self, name, mode="r"):
            z = np.load(os.path.splitext(str(name))[0] + ".npz")
            super().__init__({k: z[k] for k in z.files})
    h5.File = File
    sys.modules["h5py"] = h5
    # the reference's datasets/ has no __init__.py (a namespace package loses to the installed `datasets` distribution):
    # load its two files by path
    import importlib.util
    def load(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, @REF@ + "/datasets/" + name + ".py")
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
    ScanObjectNN = load("scanobjectnn").ScanObjectNN
    Indoor3DSemSeg = load("s3dis_v2").Indoor3DSemSeg
else:
    sys.path.insert(0, @ROOT@)
    from datasets.scanobjectnn import ScanObjectNN
    from datasets.s3dis_v2 import Indoor3DSemSeg
    import datasets.scanobjectnn as M
    assert @ROOT@ in M.__file__, M.__file__
out = {}
for train in (False, True):
    ds = ScanObjectNN(os.path.join(d, "scan.h5"), train=train, subsample=64 if train else None)
    np.random.seed(3); random.seed(3)
    for i in range(len(ds)):
        pc, lab, ma = ds[i]
        out["scan_%d_%d_pc" % (train, i)] = pc.numpy(); out["scan_%d_%d_ma" % (train, i)] = ma.numpy(); out["scan_%d_%d_l" % (train, i)] = np.asarray(lab)
for train, aug in ((True, True), (False, False), (True, False)):
    ds = Indoor3DSemSeg(d, 96, train=train, aug=aug)
    out["s3_len_%d" % train] = np.asarray(len(ds))
    np.random.seed(5); random.seed(5)
    for rep in range(3):                       # several passes: the 20 % / 95 % branches of the chromatic transforms
        for i in range(len(ds)):
            p, l = ds[i]
            out["s3_%d%d_%d_%d_p" % (train, aug, rep, i)] = p.numpy(); out["s3_%d%d_%d_%d_l" % (train, aug, rep, i)] = l.numpy()
np.savez(os.path.join(d, side + "_items.npz"), **out)
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (build container only)")
def test_items_equal_the_reference_loaders(tmp_path):
    d = str(tmp_path)
    make_files(d)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    for side in ("ref", "ours"):
        r = subprocess.run([sys.executable, "-c", CHILD.replace("@REF@", repr(REF)).replace("@ROOT@", repr(ROOT)), side, d], capture_output=True, text=True,
                           timeout=600, env=env, cwd="/tmp")
        assert r.returncode == 0, (side, r.stderr[-3000:])
    a, b = np.load(os.path.join(d, "ref_items.npz")), np.load(os.path.join(d, "ours_items.npz"))
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 100
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert np.allclose(a[k], b[k], rtol=0, atol=1e-6), (k, float(np.abs(a[k].astype(np.float64) - b[k]).max()))


def test_missing_reader_is_reported(tmp_path):
    from cloud_transformers_amd.data.datasets import read_arrays
    with pytest.raises((ImportError, OSError)):
        read_arrays(str(tmp_path / "absent.h5"), ("data",))

"""Dataset classes of the reference's ScanObjectNN and S3DIS (1x1 m blocks) loaders: same constructor arguments, same
items, the random augmentations consume numpy's / python's global generators in the reference's order, so a seeded run
yields the reference's samples (datasets/scanobjectnn.py:87-125, datasets/s3dis_v2.py:494-574 and the transforms they call:
scanobjectnn.py:9-84, s3dis_v2.py:32-54,82-95,117-129,185-196,246-366).

The files are HDF5.  `h5py` is a third-party reader that this image does not ship: `read_arrays` uses it when importable and
otherwise accepts the same arrays as `<file>.npz` (or `<stem>.npz`) next to the .h5 — `tools/h5_to_npz.py` converts on a
machine that has h5py.  Nothing here touches the GPU."""
import os
import pathlib
import random

import numpy as np
import torch
from torch.utils.data import Dataset


def read_arrays(path, keys):
    """{key: ndarray} of an HDF5 file (h5py), or of its .npz twin when h5py is not installed"""
    path = str(path)
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None and os.path.exists(path):
        with h5py.File(path, "r") as f:
            return {k: f[k][:] for k in keys}
    for twin in (path + ".npz", os.path.splitext(path)[0] + ".npz"):
        if os.path.exists(twin):
            with np.load(twin) as f:
                return {k: f[k] for k in keys}
    raise ImportError("reading %s needs h5py (not installed here) or a converted %s.npz beside it" % (path, os.path.splitext(path)[0]))


# ---------------------------------------------------------------------------
# ScanObjectNN (datasets/scanobjectnn.py)
# ---------------------------------------------------------------------------
def rotate_point_cloud(batch_data):
    """one random rotation about the up (y) axis per cloud; [B,N,3] -> float32 [B,N,3]"""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        a = np.random.uniform() * 2 * np.pi
        c, s = np.cos(a), np.sin(a)
        out[k] = batch_data[k].reshape(-1, 3) @ np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return out


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05):
    assert clip > 0
    B, N, C = batch_data.shape
    return np.clip(sigma * np.random.randn(B, N, C), -clip, clip) + batch_data


def center_data(pcs):
    for pc in pcs:
        pc -= pc.mean(axis=0)
    return pcs


def normalize_data(pcs):
    for pc in pcs:
        pc /= np.sqrt((np.abs(pc) ** 2).sum(axis=-1)).max()
    return pcs


def convert_to_binary_mask(masks):
    """background points carry -1 in the file's mask: 0 there, 1 elsewhere (float64, like np.ones)"""
    return np.where(masks == -1, 0.0, 1.0)


def load_withmask_h5(h5_filename):
    a = read_arrays(h5_filename, ("data", "label", "mask"))
    return a["data"], a["label"], a["mask"]


class ScanObjectNN(Dataset):
    def __init__(self, data_dir, center=True, normalize=True, train=False, subsample=None):
        self.data, self.label, self.mask = load_withmask_h5(data_dir)
        self.mask = convert_to_binary_mask(self.mask)
        if center:
            self.data = center_data(self.data)
        if normalize:
            self.data = normalize_data(self.data)
        self.train = train
        self.subsample = subsample

    def __getitem__(self, item):
        cloud = self.data[item][None]
        if self.train:
            cloud = rotate_point_cloud(jitter_point_cloud(cloud))
        pc, ma = cloud[0].copy(), self.mask[item].copy()
        if self.subsample is not None:
            idx = np.random.choice(pc.shape[0], size=self.subsample, replace=False)
            pc, ma = pc[idx], ma[idx]
        return torch.from_numpy(pc).type(torch.FloatTensor), self.label[item], torch.from_numpy(ma).type(torch.LongTensor)

    def __len__(self):
        return self.data.shape[0]


# ---------------------------------------------------------------------------
# S3DIS 1x1 m blocks (datasets/s3dis_v2.py): rows are [x, y, z, r, g, b, ...] with colours in [0, 1]
# ---------------------------------------------------------------------------
class RandomRotate:
    def __init__(self, rotate_angle=None, along_z=True):
        self.rotate_angle, self.along_z = rotate_angle, along_z

    def __call__(self, data):
        a = np.random.uniform() * 2 * np.pi if self.rotate_angle is None else self.rotate_angle
        c, s = np.cos(a), np.sin(a)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) if self.along_z else np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        data[:, 0:3] = data[:, 0:3] @ R.T
        return data


class RandomScale:
    def __init__(self, scale_low=0.8, scale_high=1.2, anisotropic=True):
        self.scale_low, self.scale_high, self.anisotropic = scale_low, scale_high, anisotropic

    def __call__(self, data):
        data[:, 0:3] *= np.random.uniform(self.scale_low, self.scale_high, size=3 if self.anisotropic else None)
        return data


class RandomSymmetries:
    def __init__(self, do_sym=(True, False, False)):
        assert len(do_sym) == 3
        self.do_sym = do_sym

    def __call__(self, data):
        signs = [np.round(np.random.uniform()) * 2 - 1 if flag else 1 for flag in self.do_sym]
        data[:, 0:3] *= np.asarray(signs, dtype=np.float32)
        return data


class RandomJitter:
    def __init__(self, sigma=0.01, clip=0.05):
        self.sigma, self.clip = sigma, clip

    def __call__(self, data):
        assert self.clip > 0
        data[:, 0:3] += np.clip(self.sigma * np.random.randn(data.shape[0], 3), -self.clip, self.clip)
        return data


class ChromaticAutoContrast:
    def __init__(self, randomize_blend_factor=True, blend_factor=0.5):
        self.randomize_blend_factor, self.blend_factor = randomize_blend_factor, blend_factor

    def __call__(self, data):
        if random.random() < 0.2:
            rgb = data[:, 3:6]
            lo, hi = rgb.min(0, keepdims=True), rgb.max(0, keepdims=True)
            stretched = (rgb - lo) * (1 / (hi - lo))
            w = random.random() if self.randomize_blend_factor else self.blend_factor
            data[:, 3:6] = (1 - w) * rgb + w * stretched
        return data


class ChromaticTranslation:
    def __init__(self, trans_range_ratio=1e-1):
        self.trans_range_ratio = trans_range_ratio

    def __call__(self, data):
        if random.random() < 0.95:
            shift = (np.random.rand(1, 3) - 0.5) * 2 * self.trans_range_ratio
            data[:, 3:6] = np.clip(shift + data[:, 3:6], 0, 1.0)
        return data


class ChromaticJitter:
    def __init__(self, std=0.01):
        self.std = std

    def __call__(self, data):
        if random.random() < 0.95:
            data[:, 3:6] = np.clip(np.random.randn(data.shape[0], 3) * self.std + data[:, 3:6], 0, 1)
        return data


class HueSaturationTranslation:
    """random hue shift and saturation scale through HSV (colorsys formulas, vectorised), colours quantised to 8 bits on
    the way back as in the reference (s3dis_v2.py:350,362)"""

    def __init__(self, hue_max, saturation_max):
        self.hue_max, self.saturation_max = hue_max, saturation_max

    @staticmethod
    def rgb_to_hsv(rgb):
        rgb = rgb.astype("float")
        r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
        hi, lo = rgb[..., :3].max(-1), rgb[..., :3].min(-1)
        span = hi - lo
        grey = span == 0
        safe = np.where(grey, 1.0, span)
        rc, gc, bc = [np.where(grey, 0.0, (hi - ch) / safe) for ch in (r, g, b)]
        h = np.select([r == hi, g == hi], [bc - gc, 2.0 + rc - bc], default=4.0 + gc - rc)
        hsv = np.zeros_like(rgb)
        hsv[..., 0] = (h / 6.0) % 1.0
        hsv[..., 1] = np.where(grey, 0.0, span / np.where(hi == 0, 1.0, hi))
        hsv[..., 2] = hi
        hsv[..., 3:] = rgb[..., 3:]
        return hsv

    @staticmethod
    def hsv_to_rgb(hsv):
        h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
        sector = (h * 6.0).astype("uint8")
        f = h * 6.0 - sector
        p, q, t = v * (1.0 - s), v * (1.0 - s * f), v * (1.0 - s * (1.0 - f))
        sector = sector % 6
        cond = [s == 0.0, sector == 1, sector == 2, sector == 3, sector == 4, sector == 5]
        rgb = np.empty_like(hsv)
        rgb[..., 0] = np.select(cond, [v, q, p, p, t, v], default=v)
        rgb[..., 1] = np.select(cond, [v, v, v, q, p, p], default=t)
        rgb[..., 2] = np.select(cond, [v, p, t, v, v, q], default=p)
        rgb[..., 3:] = hsv[..., 3:]
        return rgb.astype("uint8")

    def __call__(self, data):
        feats = data[:, 3:6] * 255.0
        hsv = self.rgb_to_hsv(feats[:, :3])
        hue = (random.random() - 0.5) * 2 * self.hue_max
        sat = 1 + (random.random() - 0.5) * 2 * self.saturation_max
        hsv[..., 0] = np.remainder(hue + hsv[..., 0] + 1, 1)
        hsv[..., 1] = np.clip(sat * hsv[..., 1], 0, 1)
        feats[:, :3] = np.clip(self.hsv_to_rgb(hsv), 0, 255)
        data[:, 3:6] = feats / 255.0
        return data


def _lines(path):
    with open(path) as f:
        return [line.rstrip() for line in f]


class Indoor3DSemSeg(Dataset):
    """indoor3d_sem_seg_hdf5_data: `all_files.txt` lists the .h5 shards (data [M,4096,9], label [M,4096]), `room_filelist.txt`
    names the room of every block; blocks of `test_area` form the test split."""

    def __init__(self, data_dir, num_points, train=True, data_precent=1.0, aug=False, test_area="Area_5"):
        super().__init__()
        self.data_precent, self.aug, self.train, self.num_points, self.test_area = data_precent, aug, train, num_points, test_area
        self.data_dir = pathlib.Path(data_dir)
        shards = [read_arrays(self.data_dir.joinpath(pathlib.Path(f).name), ("data", "label"))
                  for f in _lines(self.data_dir.joinpath("all_files.txt"))]
        points = np.concatenate([s["data"] for s in shards], 0)
        labels = np.concatenate([s["label"] for s in shards], 0)
        rooms = _lines(self.data_dir.joinpath("room_filelist.txt"))
        keep = [i for i, room in enumerate(rooms) if (self.test_area in room) != bool(self.train)]
        self.points, self.labels = points[keep, ...], labels[keep, ...]

    def __getitem__(self, idx):
        order = np.arange(0, self.num_points)
        np.random.shuffle(order)
        pts = self.points[idx, order, :6].copy()
        if self.aug:
            for t in (RandomRotate(along_z=True), RandomScale(anisotropic=True), RandomSymmetries(), RandomJitter(),
                      ChromaticAutoContrast(), ChromaticTranslation(0.10), ChromaticJitter(0.05), HueSaturationTranslation(0.5, 0.20)):
                pts = t(pts)
        return (torch.from_numpy(pts).type(torch.FloatTensor),
                torch.from_numpy(self.labels[idx, order].copy()).type(torch.LongTensor))

    def __len__(self):
        return int(self.points.shape[0] * self.data_precent)

    def set_num_points(self, pts):
        self.num_points = pts

    def randomize(self):
        pass

"""Evaluation metrics on the GPU: F-score@th and GRNet-style Chamfer distance.

Counterpart of the reference's `utils/f1_metric.py` (calculate_fscore :9-30,
get_f1_scores :33-57, get_f1_scores_merge :68-90) and `utils/grdnet_utils.py`
(ChamferDistance :9-23, AverageMeter :26-65, Metrics :68-129).  The reference
moves every cloud to the host and asks open3d's KD-tree for the nearest-neighbour
distances, one cloud at a time; here both directions of the nearest-neighbour
search for the whole batch are one ct_chamfer_fwd launch (the same kernel the
Chamfer loss uses), the threshold counts stay on the device, and there is a single
device->host copy of the [B, 3] result.

open3d returns Euclidean (not squared) distances in float64; the kernel returns
squared float32 distances, so the comparison is `sqrt(d2) < th` in float32.  Only a
distance within one float32 ulp of the threshold can be classified differently.
"""
import torch

from .chamfer import ChamferFunction


def nn_distances(xyz1, xyz2):
    """Euclidean distance from every point of xyz1 [B,n,3] to its nearest neighbour in
    xyz2 [B,m,3] and vice versa -> ([B,n], [B,m]).  No gradient (evaluation only)."""
    with torch.no_grad():
        d1, d2 = ChamferFunction.apply(xyz1.contiguous().float(), xyz2.contiguous().float())
        return torch.sqrt(d1), torch.sqrt(d2)


def fscore_batch(gt, pr, th=0.01):
    """F-score of calculate_fscore(gt, pr, th) (f1_metric.py:9-30) for a batch of clouds
    gt [B,n,3], pr [B,m,3] -> float32 tensor [B,3] = (fscore, precision, recall) on the device.

    As in the reference: d1 = gt->pr distances, d2 = pr->gt distances,
    recall = |d2 < th| / |d2|, precision = |d1 < th| / |d1| (f1_metric.py:14-19), and
    fscore = 0 when recall + precision == 0 or either cloud is empty (:21-28)."""
    B = gt.size(0)
    if gt.size(1) == 0 or pr.size(1) == 0:
        return torch.zeros(B, 3, device=gt.device)
    d1, d2 = nn_distances(gt, pr)
    precision = (d1 < th).float().mean(dim=1)
    recall = (d2 < th).float().mean(dim=1)
    s = recall + precision
    fscore = torch.where(s > 0, 2 * recall * precision / s.clamp_min(1e-30), torch.zeros_like(s))
    return torch.stack([fscore, precision, recall], dim=1)


def calculate_fscore(gt, pr, th=0.01):
    """Single pair of clouds gt [n,3], pr [m,3] -> (fscore, precision, recall) python floats."""
    f, p, r = fscore_batch(gt[None], pr[None], th)[0].tolist()
    return f, p, r


def get_f1_scores(pcd, pcd_gt, th=0.01):
    """pcd [B,3,N], pcd_gt [B,3,M] (channels first, as the generators return them) ->
    (fs, ps, rs) lists of B floats (f1_metric.py:33-57)."""
    assert pcd.shape[0] == pcd_gt.shape[0]
    res = fscore_batch(pcd_gt.detach().permute(0, 2, 1), pcd.detach().permute(0, 2, 1), th).cpu()
    return res[:, 0].tolist(), res[:, 1].tolist(), res[:, 2].tolist()


def resample_pcd(pcd, n, generator=None):
    """Drop or duplicate rows of pcd [P, ...] so that exactly n remain (f1_metric.py:60-65 /
    pcd_utils.py:16-21): a random permutation, topped up with uniformly drawn repeats."""
    P = pcd.shape[0]
    idx = torch.randperm(P, device=pcd.device, generator=generator)
    if P < n:
        idx = torch.cat([idx, torch.randint(P, (n - P,), device=pcd.device, generator=generator)])
    return pcd[idx[:n]]


def get_f1_scores_merge(pcd, pcd_2, pcd_gt, th=0.01, generator=None):
    """Two predictions [B,3,N1], [B,3,N2] are concatenated and resampled to the ground truth's
    point count before scoring (f1_metric.py:68-90)."""
    assert pcd.shape[0] == pcd_gt.shape[0]
    assert pcd.shape[0] == pcd_2.shape[0]
    merged = torch.cat([pcd, pcd_2], dim=-1).detach().permute(0, 2, 1)      # [B, N1+N2, 3]
    n = pcd_gt.shape[-1]
    merged = torch.stack([resample_pcd(merged[b], n, generator) for b in range(merged.shape[0])])
    res = fscore_batch(pcd_gt.detach().permute(0, 2, 1), merged, th).cpu()
    return res[:, 0].tolist(), res[:, 1].tolist(), res[:, 2].tolist()


class ChamferDistance(torch.nn.Module):
    """mean(dist1) + mean(dist2) on [B,n,3] clouds; with `ignore_zeros` and batch size 1, points
    whose coordinates sum to zero (padding) are dropped first (grdnet_utils.py:9-23)."""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(dim=0)
            xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(dim=0)
        dist1, dist2 = ChamferFunction.apply(xyz1.contiguous(), xyz2.contiguous())
        return torch.mean(dist1) + torch.mean(dist2)


class AverageMeter(object):
    """Running mean of one scalar, or of a row of named scalars (the interface the completion eval loop uses:
    `AverageMeter(names)`, `.update(value | [values])`, `.val()/.count()/.avg()` with an optional column index —
    grdnet_utils.py:26-65).  Kept as one running record per column."""

    class _Column(object):
        __slots__ = ("last", "total", "n")

        def __init__(self):
            self.last, self.total, self.n = 0, 0, 0

        def push(self, v):
            self.last = v
            self.total += v
            self.n += 1

        def mean(self):
            return self.total / self.n if self.n else 0.0

    def __init__(self, items=None):
        self.items = items
        self.n_items = len(items) if items is not None else 1
        self.reset()

    def reset(self):
        self._cols = [AverageMeter._Column() for _ in range(self.n_items)]

    def update(self, values):
        row = values if isinstance(values, list) else [values]
        for col, v in zip(self._cols, row):
            col.push(v)

    def _read(self, field, idx):
        if idx is not None:
            return field(self._cols[idx])
        if self.items is None:
            return field(self._cols[0])
        return [field(c) for c in self._cols]

    def val(self, idx=None):
        return self._read(lambda c: c.last, idx)

    def count(self, idx=None):
        return self._read(lambda c: c.n, idx)

    def avg(self, idx=None):
        return self._read(lambda c: c.mean(), idx)


class Metrics(object):
    """The completion benchmark's metric pair: F-Score@0.01 and Chamfer distance x1000
    (grdnet_utils.py:68-129).  `Metrics.get(pred [1,n,3], gt [1,m,3]) -> [fscore, cd1000]`."""

    NAMES = ["F-Score", "ChamferDistance"]
    GREATER_IS_BETTER = {"F-Score": True, "ChamferDistance": False}
    INIT_VALUE = {"F-Score": 0, "ChamferDistance": 32767}
    _chamfer = ChamferDistance(ignore_zeros=True)

    @classmethod
    def names(cls):
        return list(cls.NAMES)

    @classmethod
    def get(cls, pred, gt):
        return [cls._get_f_score(pred, gt), cls._get_chamfer_distance(pred, gt)]

    @classmethod
    def _get_f_score(cls, pred, gt, th=0.01):
        # here dist1 = pred->gt gives precision and dist2 = gt->pred gives recall (grdnet_utils.py:113-116)
        pred = pred.reshape(1, -1, 3)
        gt = gt.reshape(1, -1, 3)
        return fscore_batch(pred, gt, th)[0, 0].item()

    @classmethod
    def _get_chamfer_distance(cls, pred, gt):
        return cls._chamfer(pred, gt).item() * 1000

    def __init__(self, metric_name, values):
        assert metric_name in self.NAMES, "Invalid metric name to compare."
        self.metric_name = metric_name
        self._values = [self.INIT_VALUE[n] for n in self.NAMES]
        if isinstance(values, list):
            self._values = values
        elif isinstance(values, dict):
            for k, v in values.items():
                if k in self.NAMES:
                    self._values[self.NAMES.index(k)] = v
        else:
            raise Exception("Unsupported value type: %s" % type(values))

    def state_dict(self):
        return dict(zip(self.NAMES, self._values))

    def __repr__(self):
        return str(self.state_dict())

    def better_than(self, other):
        if other is None:
            return True
        i = self.NAMES.index(self.metric_name)
        mine, theirs = self._values[i], other._values[i]
        return mine > theirs if self.GREATER_IS_BETTER[self.metric_name] else mine < theirs


def sphere_noise(batch, num_pts, device, generator=None):
    """Uniform samples on the unit sphere, [batch, 3, num_pts] (pcd_utils.py:5-13):
    theta ~ U(0, 2pi), phi = acos(1 - 2u)."""
    with torch.no_grad():
        u = torch.rand(2, batch, num_pts, device=device, generator=generator)
        theta = (2 * torch.pi) * u[0]
        phi = torch.acos(1 - 2 * u[1])
        sin_phi = torch.sin(phi)
        return torch.stack([sin_phi * torch.cos(theta), sin_phi * torch.sin(theta), torch.cos(phi)], dim=1)

"""GPU evaluation metrics (F-score@th, Chamfer x1000) against the KD-tree oracle
(oracle/ref_cpu.py:fscore, following the reference's utils/f1_metric.py:9-30)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _clouds(B, n, m, seed, noise=0.01):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(B, 3, m, generator=g)
    take = torch.randint(m, (B, n), generator=g)
    pr = torch.gather(gt, 2, take[:, None].expand(B, 3, n)) + noise * torch.randn(B, 3, n, generator=g)
    return pr, gt


@pytest.mark.parametrize("B,n,m,th", [(2, 1024, 1024, 0.01), (3, 777, 2048, 0.02), (1, 4096, 1000, 0.005)])
def test_f1_scores_match_kdtree_oracle(B, n, m, th):
    from cloud_transformers_amd.metrics import get_f1_scores
    pr, gt = _clouds(B, n, m, seed=B * 100 + n)
    fs, ps, rs = get_f1_scores(pr.cuda(), gt.cuda(), th)
    assert len(fs) == len(ps) == len(rs) == B
    for b in range(B):
        f, p, r = R.fscore(gt[b].T.numpy(), pr[b].T.numpy(), th)
        # a distance within float32 rounding of the threshold may flip one point either way
        assert abs(ps[b] - p) <= 1.5 / m and abs(rs[b] - r) <= 1.5 / n
        assert abs(fs[b] - f) <= 3.0 / min(n, m)
        assert 0.05 < f < 1.0          # the case actually exercises the threshold


def test_fscore_known_answers():
    from cloud_transformers_amd.metrics import calculate_fscore, fscore_batch
    gt = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]).cuda()
    pr = torch.tensor([[0., 0, 0.005], [1, 0, 0.02], [5, 5, 5]]).cuda()
    f, p, r = calculate_fscore(gt, pr, th=0.01)
    # gt->pr: only gt[0] is within 0.01 -> precision 1/4; pr->gt: only pr[0] -> recall 1/3
    assert p == pytest.approx(0.25) and r == pytest.approx(1 / 3)
    assert f == pytest.approx(2 * 0.25 / 3 / (0.25 + 1 / 3))
    assert calculate_fscore(gt, gt.clone(), 0.01) == (1.0, 1.0, 1.0)
    assert calculate_fscore(gt, gt + 10.0, 0.01) == (0.0, 0.0, 0.0)       # recall + precision == 0
    assert fscore_batch(gt[None], gt[None, :0]).tolist() == [[0.0, 0.0, 0.0]]   # empty cloud


def test_merge_resamples_to_gt_size():
    from cloud_transformers_amd.metrics import get_f1_scores_merge, resample_pcd
    pr, gt = _clouds(2, 512, 1024, seed=5)
    g = torch.Generator(device="cuda").manual_seed(0)
    fs, ps, rs = get_f1_scores_merge(pr.cuda(), pr.cuda(), gt.cuda(), 0.02, generator=g)
    assert len(fs) == 2 and all(0 < f <= 1 for f in fs)
    x = torch.arange(10, device="cuda")[:, None].float()
    assert sorted(resample_pcd(x, 10)[:, 0].tolist()) == list(range(10))       # a permutation
    assert resample_pcd(x, 4).shape[0] == 4 and resample_pcd(x, 25).shape[0] == 25
    assert set(resample_pcd(x, 25)[:10, 0].tolist()) == set(range(10))         # every point kept once first


def test_grnet_metrics_pair():
    from cloud_transformers_amd.metrics import AverageMeter, ChamferDistance, Metrics
    pr, gt = _clouds(1, 2048, 2048, seed=9)
    pred = pr.permute(0, 2, 1).contiguous().cuda()
    gtc = gt.permute(0, 2, 1).contiguous().cuda()
    f, cd = Metrics.get(pred, gtc)
    fo, _, _ = R.fscore(pr[0].T.numpy(), gt[0].T.numpy(), 0.01)
    assert abs(f - fo) <= 3.0 / 2048
    d1, d2, _, _ = R.chamfer_fwd(pr.permute(0, 2, 1).contiguous(), gt.permute(0, 2, 1).contiguous())
    assert cd == pytest.approx(1000 * float(d1.mean() + d2.mean()), rel=1e-4)
    # ignore_zeros drops all-zero padding rows when the batch is one cloud
    padded = torch.cat([pred, torch.zeros(1, 100, 3, device="cuda")], dim=1)
    assert float(ChamferDistance(ignore_zeros=True)(padded, gtc)) == pytest.approx(cd / 1000, rel=1e-5)
    m = AverageMeter(Metrics.names())
    m.update([f, cd]); m.update([1.0, 0.0])
    assert m.avg() == [pytest.approx((f + 1) / 2), pytest.approx(cd / 2)] and m.count(0) == 2
    assert Metrics("F-Score", [0.5, 3.0]).better_than(Metrics("F-Score", {"F-Score": 0.4}))
    assert Metrics("ChamferDistance", [0.5, 3.0]).better_than(None)
    assert not Metrics("ChamferDistance", [0.5, 3.0]).better_than(Metrics("ChamferDistance", [0.1, 2.0]))


def test_sphere_noise_on_unit_sphere():
    from cloud_transformers_amd.metrics import sphere_noise
    x = sphere_noise(4, 8192, torch.device("cuda"))
    assert x.shape == (4, 3, 8192)
    np.testing.assert_allclose(x.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    assert abs(float(x.mean())) < 0.02 and abs(float((x[:, 2] > 0).float().mean()) - 0.5) < 0.02

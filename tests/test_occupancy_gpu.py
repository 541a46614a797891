"""The occupancy statistic of a block in one launch (ct_grid_occupancy_ratio) against torch's own arithmetic for
`count.float() / (B*C*H)` (layers/multihead_ct.py:104-105): bit for bit, launch after launch (the arrival ticket is handed back as
zero), eagerly and replayed from a HIP graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(8, 64, 128, 128), (2, 64, 32, 32, 32), (1, 3, 5, 7), (8, 512, 8, 8, 8), (1, 1, 1, 1)], ids=str)
def test_ratio_equals_count_float_div(shape):
    from cloud_transformers_amd import ops
    torch.manual_seed(3)
    z = torch.relu(torch.randn(*shape, device="cuda") - 0.5)           # about 70 % zeros, like a rasterised grid
    z.view(-1)[:3] = torch.tensor([1e-10, -1e-10, 2e-9], device="cuda")[: min(3, z.numel())]      # around the 1e-9 threshold
    K = shape[0] * shape[1] * 3
    want = (z.abs() > 1e-9).sum().float() / K
    for _ in range(3):
        got = ops.grid_occupancy_ratio(z, K)
        assert got.dtype == torch.float32 and got.dim() == 0
        assert torch.equal(got, want), (float(got), float(want))
    assert torch.equal(ops.grid_occupancy_count(z).float() / K, want)
    zo = torch.zeros(z.numel() + 1, device="cuda")[1:]                 # a view that is not 16-byte aligned
    zo.copy_(z.view(-1))
    assert torch.equal(ops.grid_occupancy_ratio(zo, K), want)


def test_ratio_replays_from_a_graph():
    from cloud_transformers_amd import ops
    z = torch.relu(torch.randn(8, 64, 64, 64, device="cuda"))
    K = 8 * 64
    ops.grid_occupancy_ratio(z, K)                                     # (workspace created outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.grid_occupancy_ratio(z, K)
    for _ in range(3):
        z.copy_(torch.relu(torch.randn_like(z)))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, (z.abs() > 1e-9).sum().float() / K)

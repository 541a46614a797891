"""`bench.py --gpus N` (N > 1, default mode) attaches the data-parallel training step to its ONE line: the shape of
that `ddp_step` object, produced here by two gloo ranks started through launch.spawn_ranks on a CPU stand-in of the segmenter
(reference wrap: train_segmentation.py:128-130; gradient averaging: utils/train_util_distributed.py:12-34)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_gloo_ranks_attach_a_ddp_step_of_the_expected_shape(tmp_path):
    from cloud_transformers_amd import launch
    dst = tmp_path / "line.json"
    rc = launch.spawn_ranks(os.path.join(ROOT, "tests", "ddp_step_probe.py"), [str(dst)], 2, timeout=240)
    assert rc == 0
    out = json.loads(dst.read_text())
    assert out["metric"] == "op-level"                 # the op-level line is kept as it was
    d = out["ddp_step"]
    assert "error" not in d, d
    for key in ("ms_per_step", "value", "collectives_per_step", "params_equal_across_ranks", "step"):
        assert key in d, key
    assert d["step"] == "eager" and d["world_size_seen"] == 2
    assert d["ms_per_step"] > 0 and abs(d["value"] - 2 * 2 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["params_equal_across_ranks"] is True, d   # averaged gradients + synchronised batch statistics: identical replicas


def test_a_failing_attachment_keeps_the_op_level_line():
    sys.path.insert(0, ROOT)
    import bench

    def boom():
        raise RuntimeError("capture refused")

    out = bench.attach_ddp_step({"value": 1.0}, boom)
    assert out["value"] == 1.0 and out["ddp_step"]["error"].startswith("RuntimeError: capture refused")


def test_capture_env_is_set_only_for_capturing_jobs():
    from cloud_transformers_amd import launch
    assert "TORCH_NCCL_ASYNC_ERROR_HANDLING" not in launch.rank_env(0, 2, 1234, base={})
    assert launch.rank_env(0, 2, 1234, base={}, capture=True)["TORCH_NCCL_ASYNC_ERROR_HANDLING"] == "0"

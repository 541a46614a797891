"""The multi-rank launcher behind `bench.py --gpus N` (cloud_transformers_amd/launch.py), at world size 2 on gloo:
ranks get the torch.distributed.run environment, rank 0's line reports the world it saw, a failing rank's exit
code comes back and the survivors are stopped."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "_launch_probe.py")


def _spawn(argv, nproc=2):
    code = ("import sys; sys.path.insert(0, %r); from cloud_transformers_amd import launch; "
            "sys.exit(launch.spawn_ranks(%r, %r, %d, timeout=90))" % (ROOT, PROBE, argv, nproc))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150)


@pytest.mark.timeout(180)
def test_spawn_two_ranks_rendezvous_and_report():
    r = _spawn([])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # only rank 0 prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2
    assert out["dt"] == 0.5 and out["sum"] == 3.0
    assert out["addr"] == "127.0.0.1" and out["local_rank"] == "0"


@pytest.mark.timeout(180)
def test_failing_rank_propagates_its_exit_code():
    r = _spawn(["fail"])
    assert r.returncode == 7


def test_launcher_detection_and_env():
    from cloud_transformers_amd import launch
    assert not launch.under_launcher({})
    assert launch.under_launcher({"RANK": "0", "WORLD_SIZE": "2"})
    env = launch.rank_env(3, 8, 1234, base={})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "1234"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus 2` without a launcher on a box with no GPU: the parent refuses before starting ranks."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)

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


def test_visible_gpus_counts_kfd_nodes_without_the_runtime(tmp_path):
    """launch.visible_gpus reads the KFD topology (nodes with SIMDs) and the *_VISIBLE_DEVICES lists — the parent of
    `bench.py --gpus N` must not initialise HIP to count devices."""
    from cloud_transformers_amd import launch
    for i, simds in enumerate([0, 256, 256, 256]):            # node 0: the CPU
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count %d\n" % simds)
    assert launch.visible_gpus({}, str(tmp_path)) == 3
    assert launch.visible_gpus({"HIP_VISIBLE_DEVICES": "0,2"}, str(tmp_path)) == 2
    assert launch.visible_gpus({"ROCR_VISIBLE_DEVICES": "1"}, str(tmp_path)) == 1
    assert launch.visible_gpus({}, str(tmp_path / "absent")) == 0


def test_concurrent_builds_serialise_on_the_lock(tmp_path, monkeypatch):
    """_lib.build() from several processes at once (ranks on a fresh checkout): one compiles, the others wait on the
    file lock and find the library built; nobody renames a file a sibling is still writing."""
    import textwrap
    script = tmp_path / "b.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        from cloud_transformers_amd import _lib
        _lib.LIB_DIR = %r
        _lib.LIB_PATH = os.path.join(_lib.LIB_DIR, "libfake.so")
        _lib.HIP_SOURCES = ["ct_chamfer.hip"]
        calls = os.path.join(_lib.LIB_DIR, "calls.%%d" %% os.getpid())
        real = _lib.subprocess.run
        def fake(cmd, check):                      # stands in for hipcc: slow, writes the -o target
            open(calls, "w").close()
            time.sleep(0.5)
            with open(cmd[-1], "w") as f:
                f.write("x")
        _lib.subprocess.run = fake
        _lib.build()
        assert os.path.exists(_lib.LIB_PATH)
    """ % (ROOT, str(tmp_path))))
    procs = [subprocess.Popen([sys.executable, str(script)]) for _ in range(3)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0, 0]
    assert len([f for f in os.listdir(tmp_path) if f.startswith("calls.")]) == 1      # one compile, two waited
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]

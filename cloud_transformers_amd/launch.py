"""One process per GPU, started from a parent that never touches the GPU.

`bench.py --gpus N` (and anything else that wants N ranks on one node) calls `spawn_ranks` BEFORE its first HIP
call: the parent only starts the children (plain `subprocess`, no exec of itself, no fork of an initialised HIP
runtime), waits for them and returns the first non-zero exit code.  Each child gets the environment
`torch.distributed.run` would give it (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT) and
initialises RCCL itself — the recipe of the reference's launcher (train_segmentation.py:58-61: one process per
device, NCCL process group, device = local rank).
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus(env=None, topology="/sys/class/kfd/kfd/topology/nodes"):
    """Number of GPUs this process may use, counted WITHOUT the HIP runtime (torch.cuda.device_count() may fall back to
    hipGetDeviceCount, which initialises HIP in the parent that is meant to stay GPU-free): KFD topology nodes that have
    SIMDs, narrowed by the first of HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES that is set.
    0 without the amdgpu compute driver (no KFD directory); None when the topology exists but cannot be read (the ranks
    then check LOCAL_RANK against their own device count)."""
    env = os.environ if env is None else env
    if not os.path.isdir(topology):
        return 0
    try:
        n = 0
        for node in os.listdir(topology):
            with open(os.path.join(topology, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
    except (OSError, ValueError, IndexError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        if env.get(var, "") != "":
            n = min(n, len([x for x in env[var].split(",") if x.strip() != ""]))
            break
    return n


def under_launcher(env=None):
    """True when this process already is one rank of a job (torch.distributed.run or spawn_ranks started it)."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def rank_env(rank, world, port, base=None, capture=False):
    """`capture`: the ranks will capture collectives into a HIP graph (harness.fit(hip_graph=True), bench.py's DDP step) — torch's
    recipe (notes/cuda.rst) then wants TORCH_NCCL_ASYNC_ERROR_HANDLING=0: the process group's watchdog otherwise queries events of
    the capturing stream and takes the process down.  The trade: with the watchdog's error handling off a hung or failed collective
    is never aborted — the job hangs instead of exiting non-zero — so eager jobs keep the default."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required by RCCL on this driver
    if capture:
        env.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    return env


def spawn_ranks(script, argv, nproc, timeout=None, python=None, capture=False):
    """Run `python script argv...` as ranks 0..nproc-1 of one job; stdout/stderr are inherited (rank 0 prints the
    result line).  Returns 0 when every rank exited 0, else the first non-zero code (the other ranks are
    terminated: a rank that died would leave them waiting in a collective)."""
    port = free_port()
    cmd = [python or sys.executable, script] + list(argv)
    procs = [subprocess.Popen(cmd, env=rank_env(r, nproc, port, capture=capture)) for r in range(nproc)]
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                break
            if live:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc

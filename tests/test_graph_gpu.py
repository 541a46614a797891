"""Whole-block HIP graph capture: every libcloudct launch (and its memsets / workspace allocations) goes to torch's
current stream, so forward + backward of a MultiHeadUnion captures into one graph and the replay reproduces the eager
gradients.  Run in a subprocess: the capture is process-global state we do not want to share with the test session."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_block_step_captures_and_replays():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "block_graph_bench.py"), "small"],
                       capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "replay reproduces the gradients: True" in r.stdout, r.stdout[-2000:]

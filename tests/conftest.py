import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """tests/golden/<name>.npz -> {case: {key: ndarray}} (keys are 'case/key')."""
    raw = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in raw.files:
        case, key = k.split("/", 1)
        out.setdefault(case, {})[key] = raw[k]
    return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get

"""Run GPU tests with the raster debug flags forced (1 = no hot kernels, 2 = hot kernels forced on small shapes):
   python tools/dev/pytest_flags.py FLAGS [pytest args...] — a robustness sweep; tests that assert kernel families are expected to differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
from cloud_transformers_amd import _lib
_lib.load().ct_debug_set_flags(int(sys.argv[1]))
sys.exit(pytest.main(sys.argv[2:]))

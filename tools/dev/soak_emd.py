"""Soak of the EMD against the oracle: tests/test_emd_gpu.py::test_fuzz_matches_oracle_exactly over many seeds (clusters, duplicated
points, shared point sets; exact equality of assignments and distances).   python3 tools/dev/soak_emd.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_emd_gpu as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(100, 100 + n):
    try:
        E.test_fuzz_matches_oracle_exactly(seed)
    except Exception as e:
        bad += 1; print("seed", seed, "FAILED", str(e)[:200].replace("\n", " "))
print("EMD fuzz:", n, "seeds, failures", bad)

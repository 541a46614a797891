"""One-off soak: many more random cases through the fuzz tests' own case generators."""
import os, sys, importlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gconv_gpu as G
bad = 0
for seed in range(100, 700):
    try:
        G.test_gconv_fuzz(seed)
    except Exception as e:   # noqa
        bad += 1
        print("gconv seed", seed, "FAILED", str(e)[:300])
print("gconv soak done, failures:", bad)

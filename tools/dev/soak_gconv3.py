"""Round-3 soak of the grouped conv: wide-group, forced C4-3D and general fuzz generators of tests/test_gconv_gpu.py over many
seeds.   python3 tools/dev/soak_gconv3.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
import tests.test_gconv_gpu as G
from cloud_transformers_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = skip = 0
for seed in range(100, 100 + n):
    try:
        G.test_gconv_fuzz_wide_groups(seed)
    except pytest.skip.Exception:
        skip += 1
    except Exception as e:
        bad += 1; print("wide seed", seed, "FAILED", str(e)[:300].replace("\n", " "))
print("wide groups:", n, "seeds, skipped", skip, "failures", bad)
bad = 0
for seed in range(100, 100 + n // 2):
    lib.ct_debug_set_gconv(4)
    try:
        G.test_gconv_fuzz_wide_groups(seed, G._rand_cfg_c4_3d, 6000)
    except pytest.skip.Exception:
        pass
    except Exception as e:
        bad += 1; print("c4 seed", seed, "FAILED", str(e)[:300].replace("\n", " "))
    finally:
        lib.ct_debug_set_gconv(0)
print("C4 3D on the matrix cores:", n // 2, "seeds, failures", bad)
bad = 0
for seed in range(100, 100 + n):
    try:
        G.test_gconv_fuzz(seed)
    except Exception as e:
        bad += 1; print("gconv seed", seed, "FAILED", str(e)[:300].replace("\n", " "))
print("general fuzz:", n, "seeds, failures", bad)

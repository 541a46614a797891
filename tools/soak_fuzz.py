"""Soak: many more random cases through the fuzz tests' own generators (tests/test_gconv_gpu.py::test_gconv_fuzz,
tests/test_raster_gpu.py::test_fuzz_shapes_against_oracle).  `python tools/soak_fuzz.py [n_gconv_seeds] [n_raster_batches] [n_emd_seeds] [n_hot_cases]`.

Round-1 runs: 600 grouped-conv seeds clean (after ct_gconv_supported: two shapes without an LDS tile plan used to fail);
672 raster cases (288, then 384 after the quad scatter kernels) with two expected differences, both the same thing: a feature that randn drew as exactly 0.0 ties with the zero floor of an empty
cell, where the oracle's scatter_reduce stand-in gives the candidate half the cotangent and torch_scatter's CPU rule (strict >)
and this implementation give it none (backward differs on exact ties only).
60 EMD fuzz seeds (sizes 1024-5120, clustered / duplicated / shared clouds): assignments and distances equal the oracle's exactly.
Round-2 run (after the hot kernels, channel-pair accumulators, 3D backward kernels): 600 grouped-conv seeds, 288 raster cases (one
expected difference of the kind above: a zero feature against the zero floor, tools/dev/fuzz_case_probe.py prints the cells),
40 EMD seeds, 800 forced-hot-kernel cases (2D/3D, max/sum, masks): clean.
Round-3 run (after the matrix-core weight gradient, the LDS-DMA bank of the K-split kernel and the C4 3D matrix-core kernel;
tools/dev/soak_gconv3.py): 300 wide-group seeds (tests/test_gconv_gpu.py::_rand_cfg_wide, 44 without a tile plan skipped),
150 forced C4 3D seeds, 300 general seeds: clean."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gconv_gpu as G          # noqa: E402
import tests.test_raster_gpu as T         # noqa: E402
import tests.test_emd_gpu as E            # noqa: E402
import tests.test_headline_gpu as HL      # noqa: E402


def hot_fuzz(n_cases, seed=7000):
    """Random shapes through the hot-shape kernels (forced on: they normally need >= 64-256 workgroups of planes): 2D and 3D,
    max and sum, padding masks, ragged quads, non-square grids — each against the oracle chain (forward, and all three
    gradients)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    mod, lib = HL._lib()
    bad = 0
    for i in range(n_cases):
        dim = int(rng.integers(2, 4))
        if dim == 2:
            W = (int(rng.choice([8, 12, 16, 24, 32, 40, 48, 64])), int(rng.choice([8, 16, 20, 32, 36, 64])))
        else:
            W = tuple(int(v) for v in rng.choice([4, 6, 8, 10, 16], size=3))
            if (W[0] * W[1] * W[2]) % 4:
                W = (W[0], W[1], 8)
        cfg = (int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.choice([4, 8, 12, 16, 20, 32])),
               4 * int(rng.integers(16, 2300)), W, bool(rng.random() < 0.4), False)
        for reduce in ("max", "sum"):
            try:
                HL.test_hot_kernels_forced_on_small_shapes(cfg, reduce, lambda v: lib.ct_debug_set_flags(v))
            except Exception as e:         # noqa: BLE001
                bad += 1
                lib.ct_debug_set_flags(0)
                print("hot cfg", cfg, reduce, "FAILED", str(e)[:300].replace("\n", " "))
    print("hot kernels:", 2 * n_cases, "cases, failures:", bad)


def main():
    n_g = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_r = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    bad = 0
    for seed in range(100, 100 + n_g):
        try:
            G.test_gconv_fuzz(seed)
        except Exception as e:             # noqa: BLE001
            bad += 1
            print("gconv seed", seed, "FAILED", str(e)[:300].replace("\n", " "))
    print("gconv:", n_g, "seeds, failures:", bad)
    bad = n = 0
    for seed in range(3000, 3000 + n_r):
        for cfg in T._fuzz_cases(n=24, seed=seed):
            n += 1
            try:
                T.test_fuzz_shapes_against_oracle(cfg)
            except Exception as e:         # noqa: BLE001
                bad += 1
                print("raster cfg", cfg, "FAILED", str(e)[:300].replace("\n", " "))
    print("raster:", n, "cases, failures:", bad)
    n_e = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = 0
    for seed in range(100, 100 + n_e):
        try:
            E.test_fuzz_matches_oracle_exactly(seed)
        except Exception as e:             # noqa: BLE001
            bad += 1
            print("emd seed", seed, "FAILED", str(e)[:300].replace("\n", " "))
    print("emd:", n_e, "seeds, failures:", bad)
    if len(sys.argv) > 4:
        hot_fuzz(int(sys.argv[4]))


if __name__ == "__main__":
    main()

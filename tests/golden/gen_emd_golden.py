"""Golden vectors for the auction EMD from THE REFERENCE'S OWN KERNELS, run on an MI355X.

oracle/_ref/emd_reference.so is the reference's emd_linear/emd_cuda.cu + emd.cpp compiled for gfx950 by `make -C oracle ref_emd`
(ROCm's hipify-perl renames two headers and three error-API calls; the kernels are the reference's, line for line).  This script
calls its `forward` exactly as the reference's Python wrapper does (emd_linear/emd_module.py:40-55: the fourteen buffers, their
dtypes and initial values) on seeded clouds and writes tests/golden/emd_reference.npz: inputs, squared distances, assignments.

The reference is racy by construction (GetMax: several bidders within 1e-6 of a target's best increment -> last writer;
the forced assignment of the last iteration: several bidders of one target -> last writer, `price +=` unsynchronised), so every
case is run RUNS times: a case whose outputs are identical in all runs is kept as a pin (`stable` = 1); an unstable one is kept
too, with the first run's outputs and stable = 0, for the record (tests pin only on the stable ones and say how many there are).

Needs a GPU:  python tests/golden/gen_emd_golden.py [out.npz]      (run on the GPU box through gpurun; the fixture is then
copied from gpurun_out/ into tests/golden/).  The .so is test infrastructure: nothing under cloud_transformers_amd/ loads it."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
RUNS = 6
# the two builds of oracle/Makefile's `ref_emd`: "strict" = -ffp-contract=off (sums of squares rounded as written: what the oracle
# and the HIP kernels evaluate, so equality is asked bit for bit), "default" = the compiler's default contraction (as nvcc's
# -fmad=true would: one valid evaluation of the same source; distances differ from the strict build by an ulp)
BUILDS = (("strict", "emd_reference_strict"), ("default", "emd_reference"))


def load_reference(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_DIR, name + ".so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_forward(ext, xyz1, xyz2, eps, iters):
    """emd_linear/emd_module.py:31-57 (emdFunction.forward), buffer for buffer."""
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    z = dict(device="cuda")
    dist = torch.zeros(B, n, **z)
    assignment = torch.zeros(B, n, dtype=torch.int32, **z) - 1
    assignment_inv = torch.zeros(B, m, dtype=torch.int32, **z) - 1
    price = torch.zeros(B, m, **z)
    bid = torch.zeros(B, n, dtype=torch.int32, **z)
    bid_increments = torch.zeros(B, n, **z)
    max_increments = torch.zeros(B, m, **z)
    unass_idx = torch.zeros(B * n, dtype=torch.int32, **z)
    max_idx = torch.zeros(B * m, dtype=torch.int32, **z)
    unass_cnt = torch.zeros(512, dtype=torch.int32, **z)
    unass_cnt_sum = torch.zeros(512, dtype=torch.int32, **z)
    cnt_tmp = torch.zeros(512, dtype=torch.int32, **z)
    rc = ext.forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments, unass_idx,
                     unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters)
    torch.cuda.synchronize()
    assert rc == 1, rc
    return dist.cpu().numpy(), assignment.cpu().numpy(), price.cpu().numpy()


def clouds(kind, B, n, seed):
    """seeded inputs in [0, 1]^3 (the reference's contract, emd_module.py:9)"""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        a, b = rng.random((B, n, 3)), rng.random((B, n, 3))
    elif kind == "clustered":          # a few tight clusters against a uniform cloud: many bidders per target early on
        c = rng.random((B, 8, 3))
        a = np.clip(c[np.arange(B)[:, None], rng.integers(0, 8, (B, n))] + 0.02 * rng.standard_normal((B, n, 3)), 0, 1)
        b = rng.random((B, n, 3))
    elif kind == "surface":            # points on a sphere against points on a cube's faces (what a completion net compares)
        v = rng.standard_normal((B, n, 3))
        a = 0.5 + 0.4 * v / np.linalg.norm(v, axis=2, keepdims=True)
        b = rng.random((B, n, 3))
        f = rng.integers(0, 3, (B, n))
        np.put_along_axis(b, f[..., None], rng.integers(0, 2, (B, n, 1)).astype(np.float64), axis=2)
    else:
        raise ValueError(kind)
    return a.astype(np.float32), b.astype(np.float32)


CASES = [  # kind, B, n, eps, iters, seed
    ("uniform", 1, 1024, 0.005, 1, 11),
    ("uniform", 2, 1024, 0.005, 3, 12),
    ("uniform", 2, 1024, 0.005, 50, 13),
    ("uniform", 1, 2048, 0.005, 50, 14),
    ("uniform", 1, 2048, 0.05, 200, 15),
    ("uniform", 2, 4096, 0.005, 20, 16),
    ("clustered", 2, 1024, 0.005, 50, 17),
    ("clustered", 1, 2048, 0.002, 30, 18),
    ("surface", 2, 1024, 0.005, 50, 19),
    ("surface", 1, 4096, 0.005, 50, 20),
    ("uniform", 1, 1024, 0.05, 1000, 21),       # converged long before the last iteration: the forced assignment is a no-op
    ("uniform", 3, 1024, 0.01, 100, 22),
    ("uniform", 4, 1024, 0.002, 30, 23),
    ("uniform", 1, 2048, 0.01, 10, 24),
    ("uniform", 1, 2048, 0.02, 100, 25),
    ("uniform", 1, 4096, 0.01, 5, 26),
    ("uniform", 1, 4096, 0.05, 60, 27),
    ("clustered", 1, 1024, 0.01, 20, 28),
    ("surface", 1, 1024, 0.02, 40, 29),
    ("uniform", 1, 3072, 0.005, 15, 30),        # n a multiple of 1024 that is not a power of two
]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "emd_reference.npz")
    assert torch.cuda.is_available(), "the reference's kernels need the MI355X"
    exts = {tag: load_reference(name) for tag, name in BUILDS}
    sys.path.insert(0, ROOT)
    from oracle import emd_ref
    from cloud_transformers_amd.emd import emdModule
    data = {"n_cases": np.int32(len(CASES)), "device": np.bytes_(torch.cuda.get_device_name(0).encode()), "runs": np.int32(RUNS)}
    for ci, (kind, B, n, eps, iters, seed) in enumerate(CASES):
        a, b = clouds(kind, B, n, seed)
        ac, bc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        st, d_or, a_or = emd_ref.forward(a, b, eps, iters)
        ties_now = emd_ref.last_getmax_ties()
        d_hip, a_hip = emdModule()(ac, bc, eps, iters)
        d_hip, a_hip = d_hip.cpu().numpy(), a_hip.cpu().numpy()
        k = "c%02d_" % ci
        data.update({k + "kind": np.bytes_(kind.encode()), k + "eps": np.float32(eps), k + "iters": np.int32(iters), k + "seed": np.int32(seed),
                     k + "xyz1": a, k + "xyz2": b})
        print("case %2d %-9s B%d n%d eps %g iters %4d:" % (ci, kind, B, n, eps, iters), flush=True)
        ties = ties_now
        data[k + "oracle_getmax_ties"] = np.int64(ties)
        print("   oracle: %d GetMax window ties (0 = the reference is deterministic on this case)" % ties, flush=True)
        for tag, _ in BUILDS:
            runs = [reference_forward(exts[tag], ac, bc, eps, iters) for _ in range(RUNS)]
            d0, a0, p0 = runs[0]
            outcomes = []          # the distinct assignments the RUNS runs produced (a deterministic case: one)
            for r in runs:
                if not any(np.array_equal(r[1], o) for o in outcomes):
                    outcomes.append(r[1])
            stable = len(outcomes) == 1 and all(np.array_equal(r[0].view(np.uint32), d0.view(np.uint32)) for r in runs)
            ulp = np.abs(d_or.view(np.int32).astype(np.int64) - d0.view(np.int32).astype(np.int64))
            same = a_or == a0
            print("   %-7s build: %d distinct outcome(s) in %d runs | oracle == run 0: assignments %s (%d differ), dist bits %s (max %d ulp where the "
                  "assignment agrees); oracle among the observed outcomes: %s | HIP == run 0: assignments %s (%d differ), dist bits %s | "
                  "distinct targets %d / %d"
                  % (tag, len(outcomes), RUNS, np.array_equal(a_or, a0), int((~same).sum()),
                     np.array_equal(d_or.view(np.uint32), d0.view(np.uint32)), int(ulp[same].max()) if same.any() else -1,
                     any(np.array_equal(a_or, o) for o in outcomes),
                     np.array_equal(a_hip, a0), int((a_hip != a0).sum()), np.array_equal(d_hip.view(np.uint32), d0.view(np.uint32)),
                     len(np.unique(a0[0])), n), flush=True)
            data.update({k + tag + "_dist": d0, k + tag + "_assignment": a0, k + tag + "_stable": np.int32(stable),
                         k + tag + "_outcomes": np.stack(outcomes)})
    np.savez_compressed(out, **data)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

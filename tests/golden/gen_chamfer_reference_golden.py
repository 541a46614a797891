"""Golden vectors for the Chamfer distance from THE REFERENCE'S OWN KERNELS (chamfer_extension/chamfer.cu + chamfer_cuda.cpp
compiled for gfx950 by `make -C oracle ref_chamfer` -> oracle/_ref/chamfer_reference.so), run on an MI355X through the call
sequence of chamfer_extension/dist_chamfer.py:12-58: seeded clouds (uniform: unique nearest neighbours almost everywhere;
lattice: every distance exact, ties abundant — the kernels scan ascending with a strict '<'), forward and backward.
-> tests/golden/chamfer_reference.npz (inputs, dist1/2, idx1/2, gradients for seeded cotangents).
Needs a GPU:  python tests/golden/gen_chamfer_reference_golden.py [out.npz]"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # kind, B, n, m, seed
    ("uniform", 2, 256, 256, 1), ("uniform", 1, 1000, 37, 2), ("uniform", 3, 77, 513, 3), ("uniform", 1, 2048, 4096, 4),
    ("lattice", 2, 300, 1003, 5), ("lattice", 1, 513, 129, 6), ("lattice", 1, 1024, 2048, 7), ("uniform", 1, 1, 1, 8),
]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "chamfer_reference.npz")
    spec = importlib.util.spec_from_file_location("chamfer_reference", os.path.join(ROOT, "oracle", "_ref", "chamfer_reference.so"))
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    data = {"n_cases": np.int32(len(CASES))}
    for ci, (kind, B, n, m, seed) in enumerate(CASES):
        g = torch.Generator().manual_seed(seed)
        if kind == "lattice":
            a = torch.randint(0, 6, (B, n, 3), generator=g).float() / 4
            b = torch.randint(0, 6, (B, m, 3), generator=g).float() / 4
        else:
            a, b = torch.rand(B, n, 3, generator=g), torch.rand(B, m, 3, generator=g)
        g1, g2 = torch.rand(B, n, generator=g), torch.rand(B, m, generator=g)
        ac, bc = a.cuda(), b.cuda()
        d1, d2 = torch.zeros(B, n, device="cuda"), torch.zeros(B, m, device="cuda")
        i1, i2 = torch.zeros(B, n, dtype=torch.int32, device="cuda"), torch.zeros(B, m, dtype=torch.int32, device="cuda")
        ext.forward(ac, bc, d1, d2, i1, i2)
        ga, gb = torch.zeros_like(ac), torch.zeros_like(bc)
        ext.backward(ac, bc, ga, gb, g1.cuda(), g2.cuda(), i1, i2)
        torch.cuda.synchronize()
        k = "c%02d_" % ci
        data.update({k + "kind": np.bytes_(kind.encode()), k + "xyz1": a.numpy(), k + "xyz2": b.numpy(), k + "g1": g1.numpy(), k + "g2": g2.numpy(),
                     k + "dist1": d1.cpu().numpy(), k + "dist2": d2.cpu().numpy(), k + "idx1": i1.cpu().numpy(), k + "idx2": i2.cpu().numpy(),
                     k + "g_xyz1": ga.cpu().numpy(), k + "g_xyz2": gb.cpu().numpy()})
        print("case %d %s B%d n%d m%d: dist1 mean %.6f" % (ci, kind, B, n, m, float(d1.mean())), flush=True)
    np.savez_compressed(out, **data)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

"""Convert the datasets' HDF5 files to the .npz twins cloud_transformers_amd.data.datasets.read_arrays accepts where h5py is
absent (run on a machine that has h5py):   python tools/h5_to_npz.py file.h5 [more.h5 ...]"""
import os
import sys

import numpy as np


def main():
    import h5py
    for path in sys.argv[1:]:
        with h5py.File(path, "r") as f:
            np.savez_compressed(os.path.splitext(path)[0] + ".npz", **{k: f[k][:] for k in f.keys()})
        print("wrote", os.path.splitext(path)[0] + ".npz")


if __name__ == "__main__":
    main()

"""Host-side data preparation of the MHCT pipelines: voxel-grid subsampling, the dataset classes of the
reference's loaders and their augmentations.  CPU only — numpy in, numpy / torch CPU tensors out."""

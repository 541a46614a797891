from cloud_transformers_amd.data.subsampling import compute  # noqa: F401

"""`utils.pcd_utils` (reference utils/pcd_utils.py:5-21)."""
from cloud_transformers_amd.metrics import resample_pcd, sphere_noise  # noqa: F401

"""`utils.grdnet_utils` (reference utils/grdnet_utils.py): ChamferDistance, AverageMeter, Metrics."""
from cloud_transformers_amd.metrics import AverageMeter, ChamferDistance, Metrics  # noqa: F401

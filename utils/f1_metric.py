"""`utils.f1_metric` (reference utils/f1_metric.py) on the GPU nearest-neighbour kernel."""
from cloud_transformers_amd.metrics import (calculate_fscore, get_f1_scores, get_f1_scores_merge,  # noqa: F401
                                            resample_pcd)

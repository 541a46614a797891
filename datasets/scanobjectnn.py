from cloud_transformers_amd.data.datasets import (ScanObjectNN, center_data, convert_to_binary_mask, jitter_point_cloud,  # noqa: F401
                                                  load_withmask_h5, normalize_data, rotate_point_cloud)

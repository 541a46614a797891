"""`utils.train_util` with the reference's names (utils/train_util.py:19-134) on this package's harness."""
from cloud_transformers_amd.harness import (worker_init_fn, get_model, check_model_paths, create_experiment,  # noqa: F401
                                            save_exp, restore_exp, restore_exp_fix, make_optimizer, make_scheduler)

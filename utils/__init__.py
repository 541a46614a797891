"""Top-level `utils` package: the reference's import paths for the pieces that sit on the
accelerated path (`utils.f1_metric`, `utils.grdnet_utils`, `utils.pcd_utils`,
`utils.train_util_distributed`).  `extend_path` keeps any other `utils/` directory on
sys.path (the reference's own `utils/train_util.py` harness) importable under the same name."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

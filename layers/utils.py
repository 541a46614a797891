from cloud_transformers_amd.layers.utils import *  # noqa: F401,F403
from cloud_transformers_amd.layers import utils as _impl
__all__ = [n for n in dir(_impl) if not n.startswith('_')]

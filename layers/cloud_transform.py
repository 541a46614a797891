from cloud_transformers_amd.layers.cloud_transform import *  # noqa: F401,F403
from cloud_transformers_amd.layers import cloud_transform as _impl
__all__ = [n for n in dir(_impl) if not n.startswith('_')]

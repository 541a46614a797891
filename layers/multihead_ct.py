from cloud_transformers_amd.layers.multihead_ct import *  # noqa: F401,F403
from cloud_transformers_amd.layers import multihead_ct as _impl
__all__ = [n for n in dir(_impl) if not n.startswith('_')]

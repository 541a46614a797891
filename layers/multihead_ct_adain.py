from cloud_transformers_amd.layers.multihead_ct_adain import *  # noqa: F401,F403
from cloud_transformers_amd.layers import multihead_ct_adain as _impl
__all__ = [n for n in dir(_impl) if not n.startswith('_')]

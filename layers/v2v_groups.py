from cloud_transformers_amd.layers.v2v_groups import *  # noqa: F401,F403
from cloud_transformers_amd.layers import v2v_groups as _impl
__all__ = [n for n in dir(_impl) if not n.startswith('_')]

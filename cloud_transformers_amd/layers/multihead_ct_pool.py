"""Reference import path `layers.multihead_ct_pool` (Splat-only pooling block)."""
from .multihead_ct import MultiHeadPool  # noqa: F401

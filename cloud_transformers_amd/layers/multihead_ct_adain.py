"""Reference import path `layers.multihead_ct_adain` (AdaIN-flavoured MHCT blocks)."""
from .multihead_ct import MultiHeadAdaIn, MultiHeadUnionAdaIn, forward_style  # noqa: F401

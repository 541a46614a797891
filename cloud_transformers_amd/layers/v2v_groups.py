"""Reference import path `layers.v2v_groups` (grouped 3D conv blocks)."""
from .grouped_conv import Basic3DBlock, Pool3DBlock, Res3DBlock, Upsample3DBlock  # noqa: F401
from .unet import EncoderDecorder, V2VModel  # noqa: F401

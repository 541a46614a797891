from cloud_transformers_amd.layers.grouped_conv import Basic2DBlock, Res2DBlock  # noqa: F401
from cloud_transformers_amd.layers.unet import DoubleConv, Down, GroupCat, OutConv, Up  # noqa: F401
import torch  # noqa: F401  (reference files do `from unet2d.unet_parts import *` and use torch / nn / F from it)
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401

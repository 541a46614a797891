"""Grouped 2D/3D convolution blocks over the rasterised planes / volumes.

Dimension-generic counterparts of the blocks the reference's model_zoo builds
its classifier / inpainter heads from: `Res2DBlock`, `Basic2DBlock`
(unet2d/unet_parts.py:9-46) and `Res3DBlock`, `Basic3DBlock`, `Pool3DBlock`,
`Upsample3DBlock` (layers/v2v_groups.py:7-70).  Attribute names (`res_branch`,
`skip_con`, `block`) and layer order inside each Sequential match, so the
reference's state dicts load with strict=True.
"""
import torch.nn as nn
import torch.nn.functional as F

from .gconv import GroupedConv2d, GroupedConv3d

# nn.Conv{2,3}d subclasses: 3^d / stride 1 / pad 1 layers run on the MFMA kernels, 1x1 skips on MIOpen
_CONV = {2: GroupedConv2d, 3: GroupedConv3d}
_CONVT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}
_BN = {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}


def _conv_bn(nd, cin, cout, k, groups):
    return [_CONV[nd](cin, cout, kernel_size=k, groups=groups, stride=1, padding=(k - 1) // 2, bias=False),
            _BN[nd](cout)]


class _BasicBlockNd(nn.Module):
    """conv(k, groups) -> BN -> ReLU"""
    nd = 2

    def __init__(self, in_planes, out_planes, kernel_size, groups=1):
        super().__init__()
        self.block = nn.Sequential(*_conv_bn(self.nd, in_planes, out_planes, kernel_size, groups),
                                   nn.ReLU(inplace=True))

    def forward(self, x):
        return self.block(x)


class _ResBlockNd(nn.Module):
    """relu( conv3-BN-ReLU-conv3-BN (x) + skip(x) ), skip = identity or 1x1 conv + BN"""
    nd = 2

    def __init__(self, in_planes, out_planes, groups=1):
        super().__init__()
        self.res_branch = nn.Sequential(*_conv_bn(self.nd, in_planes, out_planes, 3, groups),
                                        nn.ReLU(inplace=True),
                                        *_conv_bn(self.nd, out_planes, out_planes, 3, groups))
        if in_planes == out_planes:
            self.skip_con = nn.Sequential()
        else:
            self.skip_con = nn.Sequential(*_conv_bn(self.nd, in_planes, out_planes, 1, groups))

    def forward(self, x):
        return F.relu(self.res_branch(x) + self.skip_con(x), inplace=True)


class Basic2DBlock(_BasicBlockNd):
    nd = 2


class Basic3DBlock(_BasicBlockNd):
    nd = 3


class Res2DBlock(_ResBlockNd):
    nd = 2


class Res3DBlock(_ResBlockNd):
    nd = 3


class Pool3DBlock(nn.Module):
    def __init__(self, pool_size):
        super().__init__()
        self.pool_size = pool_size

    def forward(self, x):
        return F.max_pool3d(x, kernel_size=self.pool_size, stride=self.pool_size)


class Upsample3DBlock(nn.Module):
    """2x transposed-conv upsampling -> BN -> ReLU"""

    def __init__(self, in_planes, out_planes, kernel_size, stride, groups=1):
        super().__init__()
        assert kernel_size == 2 and stride == 2
        self.block = nn.Sequential(
            _CONVT[3](in_planes, out_planes, kernel_size=kernel_size, groups=groups, stride=stride,
                      padding=0, output_padding=0, bias=False),
            _BN[3](out_planes),
            nn.ReLU(inplace=True))

    def forward(self, x):
        return self.block(x)

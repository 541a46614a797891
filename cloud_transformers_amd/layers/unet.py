"""Grouped U-Net over rasterised planes and the grouped V2V encoder-decoder over volumes.

None of the reference's model_zoo files instantiate these (its heads use the Res2D / Res3D stacks), but they are part of
the import surface of `unet2d.unet_parts`, `unet2d.unet_model` (unet2d/unet_parts.py:49-150, unet2d/unet_model.py:8-41)
and `layers.v2v_groups` (layers/v2v_groups.py:73-171), so a reference-style file importing them must find them, with the
same constructor arguments, sub-module names (state-dict keys) and forward semantics.  The 3^d / stride 1 / pad 1 grouped
convolutions inside run on this package's kernels (layers/gconv.py), the rest on torch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .gconv import GroupedConv2d, GroupedConv3d
from .grouped_conv import Basic3DBlock, Pool3DBlock, Res3DBlock, Upsample3DBlock


class GroupCat(nn.Module):
    """Concatenate two grouped feature maps group by group: channels (g, f1) and (g, f2) -> (g, f1 + f2)."""

    def __init__(self, groups):
        super().__init__()
        self.groups = groups

    def forward(self, x_1, x_2):
        assert x_1.shape[0] == x_2.shape[0] and x_1.shape[2:] == x_2.shape[2:]
        B, g = x_1.shape[0], self.groups
        sp = x_1.shape[2:]
        parts = [x.reshape(B, g, x.shape[1] // g, *sp) for x in (x_1, x_2)]
        return torch.cat(parts, dim=2).reshape(B, -1, *sp)


class DoubleConv(nn.Module):
    """Two rounds of grouped 3x3 conv (with bias) -> BatchNorm -> ReLU."""

    def __init__(self, in_channels, out_channels, groups):
        super().__init__()
        self.groups = groups
        layers = []
        for cin in (in_channels, out_channels):
            layers += [GroupedConv2d(cin, out_channels, groups=groups, kernel_size=3, padding=1),
                       nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True)]
        self.double_conv = nn.Sequential(*layers)

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    """2x max-pool, then DoubleConv."""

    def __init__(self, in_channels, out_channels, groups):
        super().__init__()
        self.groups = groups
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels, groups=groups))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    """2x upsampling of x1 (bilinear, align_corners, or a grouped transposed conv), centre-padded to x2's extent,
    group-wise concatenation (x2 first), DoubleConv."""

    def __init__(self, in_channels, out_channels, groups, bilinear=True):
        super().__init__()
        self.groups = groups
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        else:
            half = in_channels // 2
            self.up = nn.ConvTranspose2d(half, half, groups=groups, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels, groups=groups)
        self.group_cat = GroupCat(groups)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(self.group_cat(x2, x1))


class OutConv(nn.Module):
    """Grouped 1x1 conv (with bias) -> BatchNorm."""

    def __init__(self, in_channels, out_channels, groups):
        super().__init__()
        self.groups = groups
        self.conv = nn.Sequential(GroupedConv2d(in_channels, out_channels, groups=groups, kernel_size=1),
                                  nn.BatchNorm2d(out_channels))

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    """Four-level grouped U-Net (16-32-64-64-64 features per group) with a global linear bottleneck whose width is fixed at
    1024 = 64 x 16 groups, as in the reference (unet2d/unet_model.py:24,35)."""

    def __init__(self, n_channels, n_out, groups, bilinear=True):
        super().__init__()
        self.n_channels, self.n_out, self.groups, self.bilinear = n_channels, n_out, groups, bilinear
        g = groups
        self.inc = DoubleConv(n_channels * g, 16 * g, g)
        widths = [16, 32, 64, 64, 64]
        for i in range(4):
            setattr(self, "down%d" % (i + 1), Down(widths[i] * g, widths[i + 1] * g, g))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.linear = nn.Linear(1024, 1024)
        ups = [(64 + 64, 64), (128, 64), (64 + 32, 32), (32 + 16, 16)]
        for i, (cin, cout) in enumerate(ups):
            setattr(self, "up%d" % (i + 1), Up(cin * g, cout * g, g, bilinear))
        self.outc = OutConv(16 * g, n_out * g, g)

    def forward(self, x):
        feats = [self.inc(x)]
        for i in range(4):
            feats.append(getattr(self, "down%d" % (i + 1))(feats[-1]))
        x = feats.pop()
        x = F.leaky_relu(x + self.linear(self.avgpool(x).reshape(-1, 1024)).reshape(-1, 1024, 1, 1), inplace=True)
        for i in range(4):
            x = getattr(self, "up%d" % (i + 1))(x, feats.pop())
        return self.outc(x)


class EncoderDecorder(nn.Module):
    """Four 2x poolings down and four transposed-conv upsamplings back, a residual skip at every scale (the class name is
    the reference's spelling: layers/v2v_groups.py:73).  Channels per group: 32 -> 32 -> 64 -> 128 -> 128."""

    _enc = [(32, 32), (32, 64), (64, 128), (128, 128)]

    def __init__(self, groups):
        super().__init__()
        self.groups = groups
        g = groups
        # creation order = the reference's (encoder, middle, decoder, skips): same seed, same initial weights
        for i, (cin, cout) in enumerate(self._enc):
            setattr(self, "encoder_pool%d" % i, Pool3DBlock(2))
            setattr(self, "encoder_res%d" % i, Res3DBlock(cin * g, cout * g, groups=g))
        self.mid_res = Res3DBlock(128 * g, 128 * g, groups=g)
        for i, (cin, cout) in reversed(list(enumerate(self._enc))):
            # the reference builds decoder_res0 WITHOUT groups (layers/v2v_groups.py:95): a dense 3^3 conv — kept
            setattr(self, "decoder_res%d" % i, Res3DBlock(cout * g, cout * g, **({"groups": g} if i else {})))
            setattr(self, "decoder_upsample%d" % i, Upsample3DBlock(cout * g, cin * g, 2, 2, groups=g))
        for i, (cin, _) in enumerate(self._enc):
            setattr(self, "skip_res%d" % i, Res3DBlock(cin * g, cin * g, groups=g))

    def forward(self, x):
        skips = []
        for i in range(4):
            skips.append(getattr(self, "skip_res%d" % i)(x))
            x = getattr(self, "encoder_res%d" % i)(getattr(self, "encoder_pool%d" % i)(x))
        x = self.mid_res(x)
        for i in (3, 2, 1, 0):
            x = getattr(self, "decoder_upsample%d" % i)(getattr(self, "decoder_res%d" % i)(x)) + skips[i]
        return x


class V2VModel(nn.Module):
    """Grouped volume-to-volume network: front (Basic + 3 Res), EncoderDecorder, back (3 Res), grouped 1x1x1 output."""

    def __init__(self, input_channels, output_channels, groups=1):
        super().__init__()
        self.groups = groups
        g = groups
        res = lambda: Res3DBlock(32 * g, 32 * g, groups=g)      # noqa: E731
        self.front_layers = nn.Sequential(Basic3DBlock(input_channels * g, 32 * g, kernel_size=3, groups=g), res(), res(), res())
        self.encoder_decoder = EncoderDecorder(groups=g)
        self.back_layers = nn.Sequential(res(), res(), res())
        self.output_layer = nn.Sequential(GroupedConv3d(32 * g, output_channels * g, groups=g, kernel_size=1, stride=1,
                                                        padding=0, bias=True))
        self._initialize_weights()

    def forward(self, x):
        return self.output_layer(self.back_layers(self.encoder_decoder(self.front_layers(x))))

    def _initialize_weights(self):
        pass

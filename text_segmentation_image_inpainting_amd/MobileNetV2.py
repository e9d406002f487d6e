"""Mirror of the partial-convolution inverted residual block (models/MobileNetV2.py:152-190).
(The MobileNetV2 / DilatedMobileNetV2 segmentation encoders of the same reference file are a
later scope row -- SURVEY.md 8(a) a12/a13.)"""
from torch import nn

from .BaseModels import BaseModule, run_nhwc, to_nchw, to_nhwc
from .masks import MaskParts, as_parts
from .partial_convolution import PartialActivatedBN, partial_convolution_block


class PartialInvertedResidual(BaseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, expansion=1, BN=True, activation=True, bias=False,
                 use_1_conv=False, no_holes_1_conv=False, same_holes=False,
                 *args, **kwargs):
        super().__init__()
        self.res_connect = stride == 1 and in_channels == out_channels          # :158
        self.conv = self.make_body(in_channels, out_channels, kernel_size, stride, padding,
                                   dilation, expansion, BN, activation, bias,
                                   use_1_conv, no_holes_1_conv, same_holes)

    @staticmethod
    def make_body(in_channels, out_channels, kernel_size, stride, padding,
                  dilation, expansion, BN, activation, bias,
                  use_1_conv, no_holes_1_conv, same_holes):
        mid_channel = int(in_channels * expansion)                              # :168
        layer = [partial_convolution_block(in_channels, mid_channel, 1, 1, 0, 1,
                                           BN=BN, activation=activation, bias=bias,
                                           use_1_conv=use_1_conv, no_holes_1_conv=no_holes_1_conv)]
        layer += [partial_convolution_block(mid_channel, mid_channel, kernel_size, stride, padding, dilation,
                                            groups=mid_channel, BN=BN, activation=activation, bias=bias,
                                            same_holes=same_holes)]
        layer += [partial_convolution_block(mid_channel, out_channels, 1, 1, 0, 1,
                                            BN=BN, activation=None, bias=bias,
                                            use_1_conv=use_1_conv, no_holes_1_conv=no_holes_1_conv)]
        return nn.Sequential(*layer)

    def forward_nhwc(self, x, mp):
        h, m = run_nhwc(self.conv[0], x, mp)
        h, m = run_nhwc(self.conv[1], h, m)
        last = self.conv[2]
        if self.res_connect and len(last) == 2 and isinstance(last[1], PartialActivatedBN):
            # residual add (:186-187) fused into the BN-apply kernel of the linear bottleneck
            h, m = last[0].forward_nhwc(h, m)
            h, m = last[1].forward_nhwc(h, m, residual=x)
            return h, m
        h, m = run_nhwc(last, h, m)
        if self.res_connect:
            h = x + h
        return h, m

    def forward(self, args):
        x, mask = args
        keep_parts = isinstance(mask, MaskParts)
        y, mp = self.forward_nhwc(to_nhwc(x), as_parts(mask))
        return to_nchw(y), (mp if keep_parts else mp.as_tensor())

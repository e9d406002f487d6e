"""Mirror of models/MobileNetV2.py: ``MobileNetV2`` / ``DilatedMobileNetV2`` encoders (:20-111, :193-216),
``InvertedResidual`` (:114-149) and the partial-convolution ``PartialInvertedResidual`` (:152-190).
(The CNN-LSTM ``MobileNetV2Classifier`` tagger of the same file is out of scope)"""
import torch
from torch import nn
from torch.utils.checkpoint import checkpoint

from . import ops
from .BaseModels import BaseModule, Conv_block, run_chain, to_nchw, to_nhwc
from .common import SpatialChannelSqueezeExcitation
from .masks import MaskParts, as_parts
from .partial_convolution import partial_convolution_block, run_block


class MobileNetV2(BaseModule):
    def __init__(self, width_mult=1, activation=nn.ReLU6(), bias=False, add_sece=False, add_partial=False,
                 image_channel=3):
        super().__init__()
        if add_partial:
            raise NotImplementedError("MobileNetV2(add_partial=True) builds inconsistent blocks in the reference "
                                      "itself (SURVEY.md F7) and is not supported")
        self.add_partial = add_partial
        self.res_block = InvertedResidual
        self.act_fn = activation
        self.bias = bias
        self.width_mult = width_mult
        self.out_stride = 32
        self.image_channel = image_channel
        self.inverted_residual_setting = [
            # t, c, n, s, dila
            [1, 16, 1, 1, 1], [6, 24, 2, 2, 1], [6, 32, 3, 2, 1], [6, 64, 4, 2, 1],
            [6, 96, 3, 1, 1], [6, 160, 3, 2, 1], [6, 320, 1, 1, 1],
        ]
        self.last_channel = 0
        self.features = self.make_inverted_resblocks(self.inverted_residual_setting, add_sece)

    def make_inverted_resblocks(self, settings, add_sece):
        in_channel = self._make_divisible(32 * self.width_mult, divisor=8)
        features = [nn.Sequential(*Conv_block(self.image_channel, in_channel, kernel_size=3, stride=2,
                                              padding=(3 - 1) // 2, bias=self.bias,
                                              BN=True, activation=self.act_fn))]
        for t, c, n, s, d in settings:
            out_channel = self._make_divisible(c * self.width_mult, divisor=8)
            block = []
            for i in range(n):
                block.append(self.res_block(in_channel, out_channel, s if i == 0 else 1, t, d,
                                            activation=self.act_fn, bias=self.bias, add_sece=add_sece))
                in_channel = out_channel
            features.append(nn.Sequential(*block))
        self.last_channel = out_channel
        return nn.Sequential(*features)

    def load_pre_train_checkpoint(self, pre_train_checkpoint, free_last_blocks):     # :71-84
        if pre_train_checkpoint:
            if isinstance(pre_train_checkpoint, str):
                self.load_state_dict(torch.load(pre_train_checkpoint, map_location="cpu"))
            else:
                self.load_state_dict(pre_train_checkpoint)
            print("Encoder check point is loaded")
        else:
            print("No check point for the encoder is loaded. ")
        if free_last_blocks >= 0:
            self.freeze_params(free_last_blocks)
        else:
            print("All layers in the encoders are re-trained. ")

    def freeze_params(self, free_last_blocks=2):                                    # :86-92
        for i in range(len(self.features) - free_last_blocks):
            for params in self.features[i].parameters():
                params.requires_grad = False
        print("{}/{} layers in the encoder are freezed.".format(len(self.features) - free_last_blocks,
                                                                len(self.features)))

    @staticmethod
    def _make_divisible(v, divisor=8, min_value=None):                             # :94-104
        if min_value is None:
            min_value = divisor
        new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
        if new_v < 0.9 * v:
            new_v += divisor
        return new_v

    def forward(self, x):
        return self.features(x)

    def forward_checkpoint(self, x):                                               # :109-111
        with self.set_activation_inplace():
            return checkpoint(self.forward, x)


class InvertedResidual(BaseModule):
    def __init__(self, in_channel, out_channel, stride, expand_ratio, dilation,
                 activation=nn.ReLU6(), bias=False, add_sece=False):
        super().__init__()
        self.stride = stride
        self.act_fn = activation
        self.bias = bias
        self.in_channels = in_channel
        self.out_channels = out_channel
        self.res_connect = self.stride == 1 and in_channel == out_channel
        self.conv = self.make_body(in_channel, out_channel, stride, expand_ratio, dilation, add_sece)

    def make_body(self, in_channel, out_channel, stride, expand_ratio, dilation, add_sece):
        mid_channel = in_channel * expand_ratio
        m = Conv_block(in_channel, mid_channel, 1, 1, 0, bias=self.bias, BN=True, activation=self.act_fn)
        m += Conv_block(mid_channel, mid_channel, 3, stride, padding=1 + (dilation - 1),
                        dilation=dilation, groups=mid_channel, bias=self.bias, BN=True, activation=self.act_fn)
        m += Conv_block(mid_channel, out_channel, 1, 1, 0, bias=self.bias, BN=True, activation=None)
        if add_sece:
            m += [SpatialChannelSqueezeExcitation(out_channel, reduction=16, activation=self.act_fn)]
        return nn.Sequential(*m)

    def forward(self, x):
        y = run_chain(list(self.conv), x)       # expand -> depth-wise -> project with the BatchNorms folded (K6b)
        if self.res_connect:
            return to_nchw(ops.add_act(to_nhwc(x), to_nhwc(y)))          # x + conv(x) (:146-147)
        return y



class DilatedMobileNetV2(MobileNetV2):
    def __init__(self, width_mult=2, activation=nn.ReLU6(), bias=False, add_sece=False, add_partial=False,
                 image_channel=3):
        super().__init__(width_mult=width_mult, activation=activation, bias=bias, add_sece=add_sece,
                         add_partial=add_partial, image_channel=image_channel)
        self.out_stride = 8
        # Rethinking Atrous Convolution for Semantic Image Segmentation                 (:203-215)
        self.inverted_residual_setting = [
            [1, 16, 1, 1, 1], [6, 24, 2, 2, 1], [6, 32, 3, 2, 1], [6, 64, 4, 1, 2],
            [6, 96, 3, 1, 4], [6, 160, 3, 1, 8], [6, 320, 1, 1, 16],
        ]
        self.features = self.make_inverted_resblocks(self.inverted_residual_setting, add_sece=add_sece)


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
        # expand -> depth-wise -> project with every BatchNorm folded into its neighbours (K6b): the two wide
        # activations are never written, the residual add (:186-187) rides in the last BatchNorm's apply kernel
        h, m = run_block(self.conv[0], x, mp, allow_lazy=True)
        h, m = run_block(self.conv[1], h, m, allow_lazy=True)
        return run_block(self.conv[2], h, m, residual=x if self.res_connect else None)

    def forward(self, args):
        x, mask = args
        keep_parts = isinstance(mask, MaskParts)
        y, mp = self.forward_nhwc(to_nhwc(x), as_parts(mask))
        return to_nchw(y), (mp if keep_parts else mp.as_tensor())

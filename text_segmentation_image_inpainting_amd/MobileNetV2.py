"""Mirror of models/MobileNetV2.py: ``MobileNetV2`` / ``DilatedMobileNetV2`` encoders (:20-111, :193-216),
``InvertedResidual`` (:114-149) and the partial-convolution ``PartialInvertedResidual`` (:152-190).
(The CNN-LSTM ``MobileNetV2Classifier`` tagger of the same file is out of scope)"""
import torch
from torch import nn

from . import ops
from .BaseModels import BaseModule, ConvSpec, build_chain, run_chain, to_nchw, to_nhwc
from .common import SpatialChannelSqueezeExcitation
from .masks import MaskParts, as_parts
from .partial_convolution import partial_convolution_block, run_block


# stage table rows: (expansion t, base channels c, blocks n, stride of the first block s, dilation d)
PLAIN_STAGES = ((1, 16, 1, 1, 1), (6, 24, 2, 2, 1), (6, 32, 3, 2, 1), (6, 64, 4, 2, 1),
                (6, 96, 3, 1, 1), (6, 160, 3, 2, 1), (6, 320, 1, 1, 1))                     # output stride 32
DILATED_STAGES = ((1, 16, 1, 1, 1), (6, 24, 2, 2, 1), (6, 32, 3, 2, 1), (6, 64, 4, 1, 2),
                  (6, 96, 3, 1, 4), (6, 160, 3, 1, 8), (6, 320, 1, 1, 16))                  # output stride 8 (:203-215)
STEM_CHANNELS = 32


def round_channels(v, divisor=8, min_value=None):
    """Channel count after the width multiplier: nearest multiple of ``divisor``, never more than 10 % below ``v``."""
    floor = divisor if min_value is None else min_value
    rounded = max(floor, int(v + divisor / 2) // divisor * divisor)
    return rounded + divisor if rounded < 0.9 * v else rounded


class InvertedResidual(BaseModule):
    """1x1 expand -> 3x3 depth-wise (stride, dilation, "same" padding) -> 1x1 linear projection [-> scSE], with the
    identity shortcut when shape-preserving (models/MobileNetV2.py:114-149)."""

    def __init__(self, in_channel, out_channel, stride, expand_ratio, dilation,
                 activation=nn.ReLU6(), bias=False, add_sece=False):
        super().__init__()
        self.act_fn, self.bias, self.stride = activation, bias, stride
        self.in_channels, self.out_channels = in_channel, out_channel
        self.res_connect = stride == 1 and in_channel == out_channel
        wide = in_channel * expand_ratio
        body, _ = build_chain(in_channel, (ConvSpec(wide, 1, bias=bias),
                                           ConvSpec(wide, 3, stride, dilation, dilation, "dw", bias=bias),
                                           ConvSpec(out_channel, 1, bias=bias, act=False)), activation)
        if add_sece:
            body.append(SpatialChannelSqueezeExcitation(out_channel, reduction=16, activation=activation))
        self.conv = nn.Sequential(*body)

    def forward(self, x):
        y = run_chain(list(self.conv), x)       # the three BatchNorms are folded into the neighbouring convs (K6b)
        return to_nchw(ops.add_act(to_nhwc(x), to_nhwc(y))) if self.res_connect else y


class MobileNetV2(BaseModule):
    """``features`` = stem conv + one nn.Sequential of InvertedResiduals per stage-table row (models/MobileNetV2.py:20-111)."""

    STAGES, OUT_STRIDE = PLAIN_STAGES, 32

    def __init__(self, width_mult=1, activation=nn.ReLU6(), bias=False, add_sece=False, add_partial=False,
                 image_channel=3):
        super().__init__()
        if add_partial:
            raise NotImplementedError("MobileNetV2(add_partial=True) builds inconsistent blocks in the reference "
                                      "itself (SURVEY.md F7) and is not supported")
        self.act_fn, self.bias, self.width_mult = activation, bias, width_mult
        self.add_partial, self.image_channel, self.out_stride = add_partial, image_channel, self.OUT_STRIDE
        self.inverted_residual_setting = [list(row) for row in self.STAGES]
        self.res_block = InvertedResidual
        width = round_channels(STEM_CHANNELS * width_mult)
        stem, _ = build_chain(image_channel, (ConvSpec(width, 3, 2, 1, bias=bias),), activation)
        stages = [nn.Sequential(*stem)]
        for t, c, n, s, d in self.STAGES:
            cout = round_channels(c * width_mult)
            stages.append(nn.Sequential(*[InvertedResidual(width if i == 0 else cout, cout, s if i == 0 else 1, t, d,
                                                           activation=activation, bias=bias, add_sece=add_sece)
                                          for i in range(n)]))
            width = cout
        self.last_channel = width
        self.features = nn.Sequential(*stages)

    _make_divisible = staticmethod(round_channels)      # the reference's name for it (:94-104)

    def load_pre_train_checkpoint(self, pre_train_checkpoint, free_last_blocks):
        """Optionally load encoder weights (a path or a state_dict), then freeze all but the last
        ``free_last_blocks`` entries of ``features`` (negative: train everything) -- stage 1 / stage 2 of the
        reference's training recipe (models/MobileNetV2.py:71-92)."""
        if pre_train_checkpoint:
            state = torch.load(pre_train_checkpoint, map_location="cpu") if isinstance(pre_train_checkpoint, str) else pre_train_checkpoint
            self.load_state_dict(state)
        if free_last_blocks >= 0:
            self.freeze_params(free_last_blocks)

    def freeze_params(self, free_last_blocks=2):
        frozen = len(self.features) - free_last_blocks
        for stage in list(self.features)[:max(frozen, 0)]:
            for p in stage.parameters():
                p.requires_grad = False
        return max(frozen, 0)

    def forward(self, x):
        return self.features(x)

    def forward_checkpoint(self, x):
        """forward() with the activations recomputed in backward (models/MobileNetV2.py:109-111): one checkpoint
        segment per ``features`` entry, so only the stage boundaries stay resident."""
        from .memory import checkpoint_sequential_stages
        with self.set_activation_inplace():
            return checkpoint_sequential_stages(list(self.features), x)


class DilatedMobileNetV2(MobileNetV2):
    """Stages 4-7 at stride 1 with dilations 2 / 4 / 8 / 16: output stride 8 (models/MobileNetV2.py:193-216)."""

    STAGES, OUT_STRIDE = DILATED_STAGES, 8

    def __init__(self, width_mult=2, activation=nn.ReLU6(), bias=False, add_sece=False, add_partial=False,
                 image_channel=3):
        super().__init__(width_mult=width_mult, activation=activation, bias=bias, add_sece=add_sece,
                         add_partial=add_partial, image_channel=image_channel)


class PartialInvertedResidual(BaseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, expansion=1, BN=True, activation=True, bias=False,
                 use_1_conv=False, no_holes_1_conv=False, same_holes=False,
                 *args, **kwargs):
        super().__init__()
        self.res_connect = stride == 1 and in_channels == out_channels
        wide = int(in_channels * expansion)
        pointwise = dict(BN=BN, bias=bias, use_1_conv=use_1_conv, no_holes_1_conv=no_holes_1_conv)
        # (cin, cout, kernel, stride, padding, dilation, extra keyword arguments) of the three partial-conv blocks (:168-179)
        rows = ((in_channels, wide, 1, 1, 0, 1, dict(pointwise, activation=activation)),
                (wide, wide, kernel_size, stride, padding, dilation,
                 dict(groups=wide, BN=BN, bias=bias, activation=activation, same_holes=same_holes)),
                (wide, out_channels, 1, 1, 0, 1, dict(pointwise, activation=None)))
        self.conv = nn.Sequential(*[partial_convolution_block(*row[:6], **row[6]) for row in rows])

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

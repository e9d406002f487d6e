"""Mirror of the two text-segmentation nets (models/text_segmentation.py:18-114): encoder (MobileNetV2 x2
with scSE / Xception) -> feature pooling (RFB / ASP) -> DeepLabV3+-style bilinear decoder.  forward() returns
logits; the caller applies the sigmoid (Examples/demo_segmentation.py:33)."""
from torch import nn

from .BaseModels import (AvgPool2d, BaseModule, Conv2d, ConvSpec, PixelShuffle, Upsample, build_chain, cat_channels, run_chain,
                         interpolate_bilinear)
from .MobileNetV2 import DilatedMobileNetV2, InvertedResidual
from .Xception import Xception
from .common import ASP, RFB

SHALLOW_STAGES = 3        # encoder.features[:3] (1/2, 1/2, 1/4) feed the decoder skip; the rest (1/8, dilated) the RFB


def _stage_width(stage):
    return stage[0].out_channels


class TextSegament(BaseModule):
    """MobileNetV2 (width 2, dilated, scSE) -> RFB over the concatenated 1/8 stages -> x2 -> concat with the pooled
    shallow stages -> two inverted residuals -> 3x3 logits conv -> x4 (models/text_segmentation.py:18-84).

    ``pixel_shuffle_head=True`` swaps the reference's head ``Conv2d(128, 1, 3) -> bilinear x4`` for
    ``Conv2d(128, 16, 3) -> PixelShuffle(4)``: the variant the reference's README describes but does not ship
     -- opt-in, performance path only, its parity is unpinned by the reference."""

    def __init__(self, encoder_checkpoint=None, free_last_blocks=-1, width_mult=2, pixel_shuffle_head=False):
        super().__init__()
        act = self.act_fn = nn.LeakyReLU(0.3)
        self.encoder = DilatedMobileNetV2(width_mult=width_mult, activation=act, bias=False, add_sece=True, add_partial=False)
        stages = list(self.encoder.features)
        shallow = sum(_stage_width(st) for st in stages[:SHALLOW_STAGES])
        deep = sum(_stage_width(st) for st in stages[SHALLOW_STAGES:])
        self.feature_avg_pool = AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.feature_pooling = RFB(deep, 256, activation=act, add_sece=True)
        block = dict(stride=1, expand_ratio=1, activation=act, add_sece=True)
        self.feature_4x_conv = InvertedResidual(shallow, 128, dilation=1, **block)
        self.smooth_feature_4x_conv = nn.Sequential(InvertedResidual(256 + 128, 128, dilation=2, **block),
                                                    InvertedResidual(128, 128, dilation=1, **block))
        if pixel_shuffle_head:
            self.out_conv = nn.Sequential(Conv2d(128, 16, kernel_size=3, padding=1, bias=True, stride=1), PixelShuffle(4))
        else:
            self.out_conv = nn.Sequential(Conv2d(128, 1, kernel_size=3, padding=1, bias=True, stride=1),
                                          Upsample(scale_factor=4, mode="bilinear", align_corners=False))
        self.initialize_weights()
        self.encoder.load_pre_train_checkpoint(encoder_checkpoint, free_last_blocks)
        # memory saver (MobileNetV2.forward_checkpoint, models/MobileNetV2.py:109-111): recompute every encoder stage in
        # backward instead of keeping its activations -- set to True for large batches / resolutions
        self.checkpoint_encoder = False

    def forward(self, x):
        stages = list(self.encoder.features)
        if self.checkpoint_encoder and self.training:
            from .memory import checkpoint_segment
            stages = [(lambda t, st=st: checkpoint_segment(st, t)) for st in stages]
        skips = []
        for stage in stages[:SHALLOW_STAGES]:
            x = stage(x)
            skips.append(x)
        # the two 1/2-resolution maps are pooled to 1/4 before the concat
        skips = [self.feature_avg_pool(t) for t in skips[:-1]] + skips[-1:]
        context = []
        for stage in stages[SHALLOW_STAGES:]:
            x = stage(x)
            context.append(x)
        pooled = interpolate_bilinear(self.feature_pooling(cat_channels(context)), 2)
        detail = self.feature_4x_conv(cat_channels(skips))
        return self.out_conv(self.smooth_feature_4x_conv(cat_channels([detail, pooled])))


class XceptionTextSegment(BaseModule):
    """Xception -> ASP(3, 5, 9) -> x2 -> concat with a 48-channel projection of the 1/4 features -> 3x3 conv ->
    3x3 logits conv -> x4 (models/text_segmentation.py:87-114)."""

    def __init__(self):
        super().__init__()
        act = self.act_fn = nn.LeakyReLU(0.3)
        self.encoder = Xception(color_channel=3, act_fn=act)
        self.feature_pooling = ASP(self.encoder.last_feature_channels, 256, act, asp_rate=(3, 5, 9))
        self.feature_4x_conv = nn.Sequential(*build_chain(self.encoder.x4_feature_channels, (ConvSpec(48, 1),), act)[0])
        head, _ = build_chain(48 + 256, (ConvSpec(128, 3, 1, 1),), act)
        self.out_conv = nn.Sequential(*head, Conv2d(128, 1, kernel_size=3, stride=1, padding=1))

    def forward(self, x):
        deep, quarter = self.encoder(x)
        pooled = interpolate_bilinear(self.feature_pooling(deep), 2)
        logits = run_chain(list(self.out_conv), cat_channels([pooled, run_chain(list(self.feature_4x_conv), quarter)]))
        return interpolate_bilinear(logits, 4)

"""Mirror of the two text-segmentation nets (models/text_segmentation.py:18-114): encoder (MobileNetV2 x2
with scSE / Xception) -> feature pooling (RFB / ASP) -> DeepLabV3+-style bilinear decoder.  forward() returns
logits; the caller applies the sigmoid (Examples/demo_segmentation.py:33)."""
from torch import nn

from .BaseModels import BaseModule, Conv2d, Conv_block, AvgPool2d, Upsample, cat_channels, interpolate_bilinear
from .MobileNetV2 import DilatedMobileNetV2, InvertedResidual
from .Xception import Xception
from .common import ASP, RFB


class TextSegament(BaseModule):
    def __init__(self, encoder_checkpoint=None, free_last_blocks=-1, width_mult=2):
        super().__init__()
        self.act_fn = nn.LeakyReLU(0.3)
        self.encoder = DilatedMobileNetV2(width_mult=width_mult, activation=self.act_fn,
                                          bias=False, add_sece=True, add_partial=False)
        self.feature_avg_pool = AvgPool2d(kernel_size=3, stride=2, padding=1)       # 1/2 -> 1/4 (:33)
        feature_channels = sum([i[0].out_channels for i in self.encoder.features[3:]])
        self.feature_pooling = RFB(feature_channels, 256, activation=self.act_fn, add_sece=True)
        concat_c = sum([i[0].out_channels for i in self.encoder.features[:3]])
        self.feature_4x_conv = InvertedResidual(concat_c, 128, stride=1, expand_ratio=1, dilation=1,
                                                activation=self.act_fn, add_sece=True)
        self.smooth_feature_4x_conv = nn.Sequential(
            InvertedResidual(256 + 128, 128, stride=1, expand_ratio=1, dilation=2, activation=self.act_fn, add_sece=True),
            InvertedResidual(128, 128, stride=1, expand_ratio=1, dilation=1, activation=self.act_fn, add_sece=True))
        self.out_conv = nn.Sequential(Conv2d(128, 1, kernel_size=3, padding=1, bias=True, stride=1),
                                      Upsample(scale_factor=4, mode="bilinear", align_corners=False))
        self.initialize_weights()
        self.encoder.load_pre_train_checkpoint(encoder_checkpoint, free_last_blocks)

    def forward(self, x):
        layer_out = []  # 1/2, 1/2, 1/4 feature maps
        for layer in self.encoder.features[:3]:
            x = layer(x)
            layer_out.append(x)
        layer_out[0] = self.feature_avg_pool(layer_out[0])
        layer_out[1] = self.feature_avg_pool(layer_out[1])
        layer_out = cat_channels(layer_out)
        pooled_features = []  # 1/8 feature maps with various dilation rates
        for layer in self.encoder.features[3:]:
            x = layer(x)
            pooled_features.append(x)
        x = self.feature_pooling(cat_channels(pooled_features))
        x = interpolate_bilinear(x, 2)
        layer_out = self.feature_4x_conv(layer_out)
        x = cat_channels([layer_out, x])
        x = self.smooth_feature_4x_conv(x)
        return self.out_conv(x)


class XceptionTextSegment(BaseModule):
    def __init__(self):
        super().__init__()
        self.act_fn = nn.LeakyReLU(0.3)
        self.encoder = Xception(color_channel=3, act_fn=self.act_fn)
        self.feature_pooling = ASP(self.encoder.last_feature_channels, 256, self.act_fn, asp_rate=(3, 5, 9))
        self.feature_4x_conv = nn.Sequential(
            *Conv_block(self.encoder.x4_feature_channels, 48, kernel_size=1, bias=False, BN=True, activation=self.act_fn))
        self.out_conv = nn.Sequential(
            *Conv_block(48 + 256, 128, kernel_size=3, stride=1, padding=1, bias=False, BN=True, activation=self.act_fn),
            Conv2d(128, 1, kernel_size=3, stride=1, padding=1))

    def forward(self, x):
        x, x4_features = self.encoder(x)
        x4_features = self.feature_4x_conv(x4_features)
        x = self.feature_pooling(x)
        x = interpolate_bilinear(x, 2)
        x = cat_channels([x, x4_features])
        x = self.out_conv(x)
        return interpolate_bilinear(x, 4)

"""Mirror of the on-path parts of models/common.py: scSE attention (:13-43), ASP (:53-93), RFB (:96-156).
(The CNN-LSTM tagger of the same file is out of scope)"""
from torch import nn

from . import ops
from .BaseModels import BaseModule, Conv2d, Conv_block, AvgPool2d, act_code, cat_channels, to_nchw, to_nhwc


class SpatialChannelSqueezeExcitation(BaseModule):
    # https://arxiv.org/abs/1709.01507 , https://arxiv.org/pdf/1803.02579v1.pdf
    def __init__(self, in_channel, reduction=16, activation=nn.ReLU()):
        super().__init__()
        linear_nodes = max(in_channel // reduction, 4)  # avoid only 1 node case        (:18)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)          # attribute kept for parity; the GAP kernel is used
        self.channel_excite = nn.Sequential(
            nn.Linear(in_channel, linear_nodes),
            activation,
            nn.Linear(linear_nodes, in_channel),
            nn.Sigmoid())
        self.spatial_excite = nn.Sequential(
            nn.Conv2d(in_channel, 1, kernel_size=1, stride=1, padding=0, bias=False),
            nn.Sigmoid())

    def forward(self, x):
        xs = to_nhwc(x)
        n, h, w, c = xs.shape
        fc1, act, fc2 = self.channel_excite[0], self.channel_excite[1], self.channel_excite[2]
        code, slope = act_code(act)
        # channel branch: GAP -> Linear -> act -> Linear -> sigmoid   (:35-37); Linear = 1x1 GEMM on [N,1,1,C]
        ch = ops.global_avg_pool(xs).view(n, 1, 1, c)
        ch = ops.pconv_pointwise(ch, fc1.weight.view(fc1.out_features, c, 1, 1), fc1.bias)
        ch = ops.activation(ch, code, slope)
        ch = ops.pconv_pointwise(ch, fc2.weight.view(c, fc1.out_features, 1, 1), fc2.bias)
        cse = ops.activation(ch, ops.ACT_SIGMOID).view(n, c)
        # spatial branch: 1x1 conv C -> 1, sigmoid                    (:41)
        sse = ops.activation(ops.pconv_pointwise(xs, self.spatial_excite[0].weight, None), ops.ACT_SIGMOID).view(n, h, w)
        return to_nchw(ops.scse_combine(xs, cse, sse))                # x*cSE + x*sSE (:38-43)


def add_SCSE_block(model_block, in_channel=None):
    if in_channel is None:
        in_channel = model_block[0].out_channels
    model_block.add_module("SCSE", SpatialChannelSqueezeExcitation(in_channel))


class ASP(BaseModule):
    # Atrous Spatial Pyramid Pooling with vortex pooling (models/common.py:53-93)
    def __init__(self, in_channel=256, out_channel=256, act_fn=None, asp_rate=(3, 9, 27)):
        super().__init__()
        self.asp = nn.Sequential(
            nn.Sequential(*Conv_block(in_channel, out_channel, kernel_size=3, stride=1, padding=1,
                                      bias=False, BN=True, activation=act_fn)),
            *[nn.Sequential(AvgPool2d(kernel_size=r, stride=1, padding=(r - 1) // 2),
                            *Conv_block(in_channel, out_channel, kernel_size=3, stride=1, padding=r,
                                        dilation=r, bias=False, BN=True, activation=act_fn)) for r in asp_rate])
        self.out_conv = nn.Sequential(*Conv_block(out_channel * 4, out_channel, kernel_size=1, bias=False,
                                                  BN=True, activation=act_fn))

    def forward(self, x):
        asp_pool = [layer(x) for layer in self.asp.children()]
        return self.out_conv(cat_channels(asp_pool))


class RFB(BaseModule):
    # receptive field block with (1xk)+(kx1) factorised convs (models/common.py:96-156)
    def __init__(self, in_channel, out_channel, activation, add_sece=False):
        super().__init__()
        asp_rate = [5, 17, 29]
        self.act_fn = activation
        self.input_down_channel = nn.Sequential(
            *Conv_block(in_channel, out_channel, kernel_size=1, bias=True, BN=True, activation=activation))
        rfb_linear_conv = [Conv2d(out_channel * 4, out_channel, kernel_size=1, bias=True)]
        if add_sece:
            rfb_linear_conv.append(SpatialChannelSqueezeExcitation(in_channel=out_channel, activation=activation))
        self.rfb_linear_conv = nn.Sequential(*rfb_linear_conv)
        self.rfb = nn.Sequential(
            self.make_pooling_branch(in_channel, out_channel, out_channel, conv_kernel=1,
                                     astro_rate=1, activation=activation, half_conv=False),
            self.make_pooling_branch(in_channel, out_channel // 2, out_channel, conv_kernel=3,
                                     astro_rate=asp_rate[0], activation=activation, half_conv=True),
            self.make_pooling_branch(in_channel, out_channel // 2, out_channel, conv_kernel=5,
                                     astro_rate=asp_rate[1], activation=activation, half_conv=True),
            self.make_pooling_branch(in_channel, out_channel // 2, out_channel, conv_kernel=7,
                                     astro_rate=asp_rate[2], activation=activation, half_conv=True))

    @staticmethod
    def make_pooling_branch(in_channel, mid_channel, out_channel, conv_kernel, astro_rate, activation, half_conv=False):
        if half_conv:
            m = nn.Sequential(
                *Conv_block(in_channel, mid_channel, kernel_size=1, padding=0,
                            bias=False, BN=True, activation=activation),
                *Conv_block(mid_channel, 3 * mid_channel // 2, kernel_size=(1, conv_kernel),
                            padding=(0, (conv_kernel - 1) // 2), bias=False, BN=True, activation=None),
                *Conv_block(3 * mid_channel // 2, out_channel, kernel_size=(conv_kernel, 1),
                            padding=((conv_kernel - 1) // 2, 0), bias=False, BN=True, activation=None),
                *Conv_block(out_channel, out_channel, kernel_size=3, dilation=astro_rate, padding=astro_rate,
                            bias=False, BN=True, activation=activation, groups=out_channel))
        else:
            m = nn.Sequential(
                *Conv_block(in_channel, out_channel, kernel_size=conv_kernel, padding=(conv_kernel - 1) // 2,
                            bias=False, BN=True, activation=activation),
                *Conv_block(out_channel, out_channel, kernel_size=3, dilation=astro_rate, padding=astro_rate,
                            bias=False, BN=True, activation=activation, groups=out_channel))
        return m

    def forward(self, x):
        rfb_pool = cat_channels([layer(x) for layer in self.rfb.children()])
        rfb_pool = self.rfb_linear_conv(rfb_pool)
        resi = self.input_down_channel(x)                                        # skip connection (:155)
        code, slope = act_code(self.act_fn)
        return to_nchw(ops.add_act(to_nhwc(rfb_pool), to_nhwc(resi), code, slope))   # act_fn(rfb_pool + resi) (:156)

"""Mirror of the on-path parts of models/common.py: scSE attention (:13-43), ASP (:53-93), RFB (:96-156).
(The CNN-LSTM tagger of the same file is out of scope)"""
from torch import nn

from . import ops
from .BaseModels import AvgPool2d, BaseModule, Conv2d, ConvSpec, act_code, build_chain, cat_channels, run_chain, to_nchw, to_nhwc


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


class ASP(BaseModule):
    """Atrous spatial pyramid with vortex pooling (models/common.py:53-93): a plain 3x3 branch plus, per rate r,
    ``AvgPool(r, stride 1) -> 3x3 dilated by r``; the four branches are concatenated and fused by a 1x1 conv."""

    def __init__(self, in_channel=256, out_channel=256, act_fn=None, asp_rate=(3, 9, 27)):
        super().__init__()
        branches = [nn.Sequential(*build_chain(in_channel, (ConvSpec(out_channel, 3, 1, 1),), act_fn)[0])]
        for r in asp_rate:
            conv, _ = build_chain(in_channel, (ConvSpec(out_channel, 3, 1, r, r),), act_fn)
            branches.append(nn.Sequential(AvgPool2d(kernel_size=r, stride=1, padding=(r - 1) // 2), *conv))
        self.asp = nn.Sequential(*branches)
        self.out_conv = nn.Sequential(*build_chain(out_channel * len(branches), (ConvSpec(out_channel, 1),), act_fn)[0])

    def forward(self, x):
        # run_chain: every conv + BatchNorm pair folded (K6b), other members (the average pools) through their own forward
        return run_chain(list(self.out_conv), cat_channels([run_chain(list(branch), x) for branch in self.asp]))


# RFB branch table: (first-conv kernel k, dilation of the closing depth-wise 3x3); k == 1: 1x1 -> dw3x3, else the
# factorised form 1x1 (to half width) -> (1 x k) -> (k x 1) -> dw3x3 (models/common.py:102,116-143)
RFB_BRANCHES = ((1, 1), (3, 5), (5, 17), (7, 29))


def rfb_branch_specs(width, k, rate):
    tail = ConvSpec(None, 3, 1, rate, rate, "dw")
    if k == 1:
        return (ConvSpec(width, 1), tail)
    half = width // 2
    return (ConvSpec(half, 1), ConvSpec(3 * half // 2, (1, k), 1, (0, (k - 1) // 2), act=False),
            ConvSpec(width, (k, 1), 1, ((k - 1) // 2, 0), act=False), tail)


class RFB(BaseModule):
    """Receptive-field block (models/common.py:96-156): four branches of growing kernel / dilation, concatenated,
    fused by a biased 1x1 conv (+ scSE), added to a 1x1 projection of the input, then the activation."""

    def __init__(self, in_channel, out_channel, activation, add_sece=False):
        super().__init__()
        self.act_fn = activation
        self.input_down_channel = nn.Sequential(*build_chain(in_channel, (ConvSpec(out_channel, 1, bias=True),), activation)[0])
        fuse = [Conv2d(out_channel * len(RFB_BRANCHES), out_channel, kernel_size=1, bias=True)]
        if add_sece:
            fuse.append(SpatialChannelSqueezeExcitation(in_channel=out_channel, activation=activation))
        self.rfb_linear_conv = nn.Sequential(*fuse)
        self.rfb = nn.Sequential(*[nn.Sequential(*build_chain(in_channel, rfb_branch_specs(out_channel, k, rate), activation)[0])
                                   for k, rate in RFB_BRANCHES])

    def forward(self, x):
        fused = self.rfb_linear_conv(cat_channels([run_chain(list(branch), x) for branch in self.rfb]))
        code, slope = act_code(self.act_fn)
        return to_nchw(ops.add_act(to_nhwc(fused), to_nhwc(self.input_down_channel(x)), code, slope))

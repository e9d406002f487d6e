"""Mirror of the slim dilated Xception encoder (models/Xception.py:13-114).  (``XceptionClassifier`` raises at
construction in the reference itself -- SURVEY.md F7 -- and is out of scope.)"""
from torch import nn

from .BaseModels import BaseModule, ConvSpec, DSConvBlock, build_chain, run_chain


class ResidualBlock(BaseModule):
    """Three depth-wise-separable convs (the last one strided, without activation) plus a shortcut: identity, or a
    strided 1x1 conv + BatchNorm when the shape changes (models/Xception.py:13-44)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, bias=False, BN=True, activation=None, expand_channel_first=True):
        super().__init__()
        mid = out_channels if expand_channel_first else in_channels
        # (cin, cout, stride, point-wise activation) of the three DSConvBlocks
        rows = ((in_channels, mid, 1, activation), (mid, out_channels, 1, activation), (out_channels, out_channels, stride, None))
        self.conv = nn.Sequential(*[DSConvBlock(ci, co, kernel_size, st, padding, dilation, bias, BN, activation, pw_act)
                                    for ci, co, st, pw_act in rows])
        self.residual_conv = None
        if stride > 1 or in_channels != out_channels:
            self.residual_conv = nn.Sequential(*build_chain(in_channels, (ConvSpec(out_channels, 1, stride, act=False),), None)[0])

    def forward(self, x):
        shortcut = x if self.residual_conv is None else run_chain(list(self.residual_conv), x)
        # the six conv + BatchNorm pairs of the three DSConvBlocks as ONE chain: every BatchNorm but the last stays virtual
        # (applied by the next conv while loading, K6b) and takes its backward reductions from that conv's dX kernel (K6c)
        mods = [m for block in self.conv for m in list(block.depth_wise_conv) + list(block.point_wise_conv)]
        # ... and the shortcut is added by the pass that writes the last BatchNorm out (x + self.conv(x), models/Xception.py:44)
        return run_chain(mods, x, final_residual=shortcut)


# flow tables: ("conv", out, stride) = 3x3 conv + BN + act; ("res", out, stride, dilation) = ResidualBlock (k 3, pad = dilation)
XCEPTION_FLOWS = (
    ("entry_flow_1", (("conv", 32, 2), ("conv", 64, 1), ("res", 128, 2, 1))),                     # -> 1/4, 128 channels
    ("entry_flow_2", (("res", 256, 2, 1), ("res", 512, 1, 2))),                                   # -> 1/8 (dilated from here)
    ("middle_flow", (("res", 512, 1, 2),) * 4 + (("res", 512, 1, 4),) * 4),
    ("exit_flow", (("res", 512, 1, 2),) * 2 + (("res", 512, 1, 1),) * 2),
)


class Xception(BaseModule):
    """Output stride 8; returns (features at 1/8 with 512 channels, features at 1/4 with 128 channels)
    (models/Xception.py:47-114)."""

    def __init__(self, color_channel=3, act_fn=nn.LeakyReLU(0.3)):
        super().__init__()
        self.act_fn = act_fn
        width = color_channel
        for name, rows in XCEPTION_FLOWS:
            mods = []
            for row in rows:
                if row[0] == "conv":
                    piece, width = build_chain(width, (ConvSpec(row[1], 3, row[2], 1),), act_fn)
                    mods += piece
                else:
                    _, cout, stride, rate = row
                    mods.append(ResidualBlock(width, cout, 3, stride=stride, padding=rate, dilation=rate, bias=False,
                                              BN=True, activation=act_fn))
                    width = cout
            setattr(self, name, nn.Sequential(*mods))
            if name == "entry_flow_1":
                self.x4_feature_channels = width
        self.last_feature_channels = width

    def forward(self, x):
        quarter = run_chain(list(self.entry_flow_1), x)       # folds the two stem conv + BatchNorm pairs (K6b)
        deep = self.exit_flow(self.middle_flow(self.entry_flow_2(quarter)))
        return deep, quarter

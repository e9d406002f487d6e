"""Mirror of the partial-convolution inpainting U-Nets (models/image_inpainting.py):
``ImageFill`` (:9-86), ``ImageFillOrigin`` (:110-191), ``DoublePartialResidual`` (:194-216) and
``ImageFillOriginV2`` (:219-290) with the reference's constructors, ``forward((x, mask)) -> x``
surface and ``state_dict`` layout.  (``ImageFillOriginV3`` crashes in the reference itself --
SURVEY.md F7 -- and is out of scope.)

forward() keeps activations NHWC and masks as planes end to end: the nearest-upsample +
concat of the decoder is one kernel (K7) and the concatenated *mask* is never materialised.
"""
from torch import nn

from . import ops
from .BaseModels import BaseModule, run_nhwc, to_nchw, to_nhwc
from .MobileNetV2 import PartialInvertedResidual
from .masks import MaskParts, Part, as_parts
from .partial_convolution import DoubleUpSample, partial_convolution_block


def _upcat_with_mask(x, mp, skip_x, skip_m):
    """DoubleUpSample + torch.cat on features and masks (models/image_inpainting.py:82-84)."""
    up_m = mp.upsample2x()
    sp = skip_m.parts[0]
    if len(skip_m.parts) != 1 or len(up_m.parts) != 1:
        raise NotImplementedError("decoder concat expects single-part masks on both sides")
    if sp.planar:
        return ops.upcat(x, skip_x), up_m.cat(skip_m)
    # general per-channel skip mask (the raw input level): multiply the skip features once here, so
    # the conv sees a row scale on the up-sampled half only; the count still uses the true mask.
    skip_pre = ops.mul_mask(skip_x, sp.full)
    part = Part(sp.channels, full=sp.full, premultiplied=True)
    part._sum = sp._sum
    return ops.upcat(x, skip_pre), MaskParts([up_m.parts[0], part])


class _UNetBase(BaseModule):
    def _encode(self, x, mp):
        fx, fm = [x], [mp]
        for layer in self.encoder:
            x, mp = run_nhwc(layer, x, mp)
            fx.append(x)
            fm.append(mp)
        return x, mp, fx[:-1], fm[:-1]

    def _decode(self, x, mp, fx, fm):
        for layer in self.decoder:
            x, mp = _upcat_with_mask(x, mp, fx.pop(-1), fm.pop(-1))
            x, mp = run_nhwc(layer, x, mp)
        return x


class ImageFill(_UNetBase):
    def __init__(self):
        super().__init__()
        self.act_fn = nn.LeakyReLU(0.3)
        self.double_upscale = DoubleUpSample(scale_factor=2, mode="nearest")
        encoder = [  # i, o, k, s, p, d, t, n
            [64, 128, 3, 2, 1, 1, 4, 2],
            [128, 256, 3, 2, 1, 1, 4, 2],
            [256, 256, 3, 2, 1, 1, 4, 2],
        ]
        self.encoder = nn.Sequential(
            partial_convolution_block(3, 64, 7, 2, 3, 1, bias=True, BN=False, activation=self.act_fn),
            *self.make_layers(encoder, use_1_conv=True, same_holes=True))
        dilated_layers = [
            [256, 256, 3, 1, 2, 2, 4, 2],
            [256, 256, 3, 1, 4, 4, 4, 2],
            [256, 256, 3, 1, 8, 8, 4, 2],
        ]
        self.dilated_layers = nn.Sequential(*self.make_layers(dilated_layers, no_holes_1_conv=True, same_holes=True))
        decoder = [
            [256 + 256, 256, 3, 1, 1, 1, 2, 1],
            [256 + 128, 128, 3, 1, 1, 1, 2, 1],
            [128 + 64, 32, 3, 1, 1, 1, 2, 1],
        ]
        self.decoder = nn.Sequential(
            *self.make_layers(decoder, no_holes_1_conv=True, same_holes=True),
            partial_convolution_block(32 + 3, 3, 3, 1, 1, 1, bias=True, BN=False, activation=False))

    def make_layers(self, settings, use_1_conv=False, no_holes_1_conv=False, same_holes=False):
        m = []
        for in_c, out_c, k, s, p, d, t, n in settings:
            layer = []
            for i in range(n):
                layer.append(PartialInvertedResidual(in_c, out_c, k, s if i == 0 else 1, p, d, t, bias=False,
                                                     BN=True, activation=self.act_fn, use_1_conv=use_1_conv,
                                                     no_holes_1_conv=no_holes_1_conv, same_holes=same_holes))
                in_c = out_c
            m.append(nn.Sequential(*layer))
        return m

    def forward(self, args):
        # mask: 1: ground truth, 0: holes
        x, mask = args
        x, mp, fx, fm = self._encode(to_nhwc(x), as_parts(mask))
        x, mp = run_nhwc(self.dilated_layers, x, mp)
        return to_nchw(self._decode(x, mp, fx, fm))


class ImageFillOrigin(_UNetBase):
    def __init__(self):
        super().__init__()
        self.double_upscale = DoubleUpSample(scale_factor=2, mode="nearest")
        encoder = [
            [64, 128, 5, 2, 2, 1, 1, 1],
            [128, 256, 5, 2, 2, 1, 1, 1],
            [256, 512, 3, 2, 1, 1, 1, 1],
            [512, 512, 3, 2, 1, 1, 1, 1],
            [512, 512, 3, 2, 1, 1, 1, 1],
            [512, 512, 3, 2, 1, 1, 1, 1],
            [512, 512, 3, 2, 1, 1, 1, 1],
        ]
        self.encoder = nn.Sequential(
            partial_convolution_block(3, 64, 7, 2, 3, 1, bias=True, BN=False, activation=nn.ReLU(), same_holes=True),
            *self.make_layer_v2(encoder, act_fn=nn.ReLU(), same_holes=True))
        decoder = [
            [512 + 512, 512, 3, 1, 1, 1, 1, 1],
            [512 + 512, 512, 3, 1, 1, 1, 1, 1],
            [512 + 512, 512, 3, 1, 1, 1, 1, 1],
            [512 + 512, 512, 3, 1, 1, 1, 1, 1],
            [512 + 256, 256, 3, 1, 1, 1, 1, 1],
            [256 + 128, 128, 3, 1, 1, 1, 1, 1],
            [128 + 64, 64, 3, 1, 1, 1, 1, 1],
        ]
        self.decoder = nn.Sequential(
            *self.make_layer_v2(decoder, act_fn=nn.LeakyReLU(0.2), same_holes=False),
            partial_convolution_block(64 + 3, 3, 3, 1, 1, 1, bias=True, BN=False, activation=False, same_holes=False))

    def make_layer_v2(self, settings, act_fn, no_holes_1_conv=False, same_holes=False):
        m = []
        for in_c, out_c, k, s, p, d, t, n in settings:
            layer = partial_convolution_block(in_c, out_c, k, s, p, d, groups=1, BN=True, activation=act_fn,
                                              bias=False, no_holes_1_conv=no_holes_1_conv, same_holes=same_holes)
            m.append(nn.Sequential(layer))
        return m

    def forward(self, args):
        x, mask = args
        x, mp, fx, fm = self._encode(to_nhwc(x), as_parts(mask))
        return to_nchw(self._decode(x, mp, fx, fm))


class DoublePartialResidual(BaseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, expansion=1, BN=True, activation=True, bias=False,
                 use_1_conv=False, no_holes_1_conv=False, same_holes=False,
                 dilation_rate=(1, 1), *args, **kwargs):
        super().__init__()
        # padding / dilation arguments are ignored, like the reference (:201-210)
        self.conv1 = partial_convolution_block(in_channels, out_channels, kernel_size, stride,
                                               padding=dilation_rate[0], dilation=dilation_rate[0],
                                               BN=BN, activation=activation, bias=bias, use_1_conv=use_1_conv,
                                               no_holes_1_conv=no_holes_1_conv, same_holes=same_holes)
        self.conv2 = partial_convolution_block(out_channels, out_channels, kernel_size, 1,
                                               padding=dilation_rate[1], dilation=dilation_rate[1],
                                               BN=BN, activation=activation, bias=bias, use_1_conv=use_1_conv,
                                               no_holes_1_conv=no_holes_1_conv, same_holes=same_holes)

    def forward_nhwc(self, x, mp):
        from .partial_convolution import PartialActivatedBN
        x1, m1 = run_nhwc(self.conv1, x, mp)
        if len(self.conv2) == 2 and isinstance(self.conv2[1], PartialActivatedBN):
            h, m2 = self.conv2[0].forward_nhwc(x1, m1)
            h, m2 = self.conv2[1].forward_nhwc(h, m2, residual=x1)               # x + out_x (:216)
            return h, m2
        x2, m2 = run_nhwc(self.conv2, x1, m1)
        return x2 + x1, m2

    def forward(self, args):
        x, mask = args
        keep_parts = isinstance(mask, MaskParts)
        y, mp = self.forward_nhwc(to_nhwc(x), as_parts(mask))
        return to_nchw(y), (mp if keep_parts else mp.as_tensor())


class ImageFillOriginV2(_UNetBase):
    def __init__(self):
        super().__init__()
        self.double_upscale = DoubleUpSample(scale_factor=2, mode="nearest")
        encoder = [
            [64, 128, 3, 2, 1, 1, 1, 1],
            [128, 256, 3, 2, 1, 1, 1, 1],
            [256, 256, 3, 2, 1, 1, 1, 1],
            [256, 256, 3, 2, 1, 1, 1, 1],
            [256, 512, 3, 2, 1, 1, 1, 1],
            [512, 512, 3, 2, 1, 1, 1, 1],
            [512, 512, 3, 2, 1, 1, 1, 1],
        ]
        self.encoder = nn.Sequential(
            partial_convolution_block(3, 64, 5, 2, 2, 1, bias=False, BN=True,
                                      activation=nn.LeakyReLU(0.2), same_holes=True),
            *self.make_layer_v2(encoder, act_fn=nn.LeakyReLU(0.2), same_holes=True, dilation_rate=(1, 2)))
        decoder = [
            [512 + 512, 512, 3, 1, 1, 1, 1, 1],
            [512 + 512, 512, 3, 1, 1, 1, 1, 1],
            [512 + 256, 256, 3, 1, 1, 1, 1, 1],
            [256 + 256, 256, 3, 1, 1, 1, 1, 1],
            [256 + 256, 256, 3, 1, 1, 1, 1, 1],
            [256 + 128, 128, 3, 1, 1, 1, 1, 1],
            [128 + 64, 64, 3, 1, 1, 1, 1, 1],
        ]
        self.decoder = nn.Sequential(
            *self.make_layer_v2(decoder, act_fn=nn.LeakyReLU(0.2), same_holes=False, dilation_rate=(2, 1)),
            partial_convolution_block(64 + 3, 3, 3, 1, 1, 1, bias=True, BN=False,
                                      activation=nn.ReLU(), same_holes=False))

    @staticmethod
    def make_layer_v2(settings, act_fn, same_holes=False, dilation_rate=(1, 1)):
        m = []
        for in_c, out_c, k, s, p, d, t, n in settings:
            layer = DoublePartialResidual(in_c, out_c, k, s, p, d, BN=True, activation=act_fn, bias=False,
                                          same_holes=same_holes, dilation_rate=dilation_rate)
            m.append(nn.Sequential(layer))
        return m

    def forward(self, args):
        x, mask = args
        x, mp, fx, fm = self._encode(to_nhwc(x), as_parts(mask))
        return to_nchw(self._decode(x, mp, fx, fm))

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
from .partial_convolution import DoubleUpSample, PartialConv, partial_convolution_block


def _upcat_with_mask(x, mp, skip_x, skip_m, virtual=False):
    """DoubleUpSample + torch.cat on features and masks (models/image_inpainting.py:82-84).  ``virtual``: the consumer can
    read the two tensors directly (ops.VirtualCat: the output head, K4c; a block opening with a 1x1 partial convolution,
    K7b) -- the concatenation is not written unless that consumer falls back to ``materialize()``."""
    up_m = mp.upsample2x()
    sp = skip_m.parts[0]
    if len(skip_m.parts) != 1 or len(up_m.parts) != 1:
        raise NotImplementedError("decoder concat expects single-part masks on both sides")
    low_plane = mp.parts[0].plane if len(mp.parts) == 1 and mp.parts[0].planar else None
    cat = (lambda a, b: ops.VirtualCat(a, b, low_plane)) if virtual else ops.upcat
    if sp.planar:
        return cat(x, skip_x), up_m.cat(skip_m)
    # general per-channel skip mask (the raw input level): multiply the skip features once here, so
    # the conv sees a row scale on the up-sampled half only; the count still uses the true mask.
    skip_pre = ops.mul_mask(skip_x, sp.full)
    part = Part(sp.channels, full=sp.full, premultiplied=True)
    part._sum = sp._sum
    return cat(x, skip_pre), MaskParts([up_m.parts[0], part])


class _UNetBase(BaseModule):
    def _encode(self, x, mp):
        fx, fm = [x], [mp]
        for layer in self.encoder:
            x, mp = run_nhwc(layer, x, mp)
            fx.append(x)
            fm.append(mp)
        return x, mp, fx[:-1], fm[:-1]

    def _decode(self, x, mp, fx, fm):
        last = len(self.decoder) - 1
        for i, layer in enumerate(self.decoder):
            # the last level feeds a bare PartialConv (the output head): hand it the concatenation unwritten; so does a level
            # whose first block is an inverted residual without a skip connection (its 1x1 expand convolution is the only reader)
            head = i == last and isinstance(layer, nn.Sequential) and len(layer) == 1 and type(layer[0]) is PartialConv
            first = layer[0] if isinstance(layer, nn.Sequential) and len(layer) > 0 else None
            expand = (isinstance(first, PartialInvertedResidual) and not first.res_connect and isinstance(first.conv[0][0], PartialConv)
                      and tuple(first.conv[0][0].feature_conv.kernel_size) == (1, 1))
            x, mp = _upcat_with_mask(x, mp, fx.pop(-1), fm.pop(-1), virtual=head or expand)
            x, mp = run_nhwc(layer, x, mp)
        return x


# Level tables.  A row is (cin, cout, kernel, stride, padding, dilation, expansion t, repeats n); decoder rows list
# cin as (up-sampled channels + skip channels).  Values: models/image_inpainting.py:15-41, :116-146, :225-252.
IMAGEFILL = dict(
    stem=(3, 64, 7, 2, 3, 1),
    encoder=((64, 128, 3, 2, 1, 1, 4, 2), (128, 256, 3, 2, 1, 1, 4, 2), (256, 256, 3, 2, 1, 1, 4, 2)),
    dilated=((256, 256, 3, 1, 2, 2, 4, 2), (256, 256, 3, 1, 4, 4, 4, 2), (256, 256, 3, 1, 8, 8, 4, 2)),
    decoder=((256 + 256, 256, 3, 1, 1, 1, 2, 1), (256 + 128, 128, 3, 1, 1, 1, 2, 1), (128 + 64, 32, 3, 1, 1, 1, 2, 1)),
    head=(32 + 3, 3, 3, 1, 1, 1))
ORIGIN = dict(
    stem=(3, 64, 7, 2, 3, 1),
    encoder=((64, 128, 5, 2, 2, 1, 1, 1), (128, 256, 5, 2, 2, 1, 1, 1), (256, 512, 3, 2, 1, 1, 1, 1)) + ((512, 512, 3, 2, 1, 1, 1, 1),) * 4,
    decoder=((512 + 512, 512, 3, 1, 1, 1, 1, 1),) * 4 + ((512 + 256, 256, 3, 1, 1, 1, 1, 1), (256 + 128, 128, 3, 1, 1, 1, 1, 1),
                                                          (128 + 64, 64, 3, 1, 1, 1, 1, 1)),
    head=(64 + 3, 3, 3, 1, 1, 1))
ORIGIN_V2 = dict(
    stem=(3, 64, 5, 2, 2, 1),
    encoder=((64, 128, 3, 2, 1, 1, 1, 1), (128, 256, 3, 2, 1, 1, 1, 1), (256, 256, 3, 2, 1, 1, 1, 1), (256, 256, 3, 2, 1, 1, 1, 1),
             (256, 512, 3, 2, 1, 1, 1, 1), (512, 512, 3, 2, 1, 1, 1, 1), (512, 512, 3, 2, 1, 1, 1, 1)),
    decoder=((512 + 512, 512, 3, 1, 1, 1, 1, 1), (512 + 512, 512, 3, 1, 1, 1, 1, 1), (512 + 256, 256, 3, 1, 1, 1, 1, 1),
             (256 + 256, 256, 3, 1, 1, 1, 1, 1), (256 + 256, 256, 3, 1, 1, 1, 1, 1), (256 + 128, 128, 3, 1, 1, 1, 1, 1),
             (128 + 64, 64, 3, 1, 1, 1, 1, 1)),
    head=(64 + 3, 3, 3, 1, 1, 1))


def build_levels(rows, make_block):
    """One ``nn.Sequential`` per table row holding its ``n`` blocks; only the first block of a level strides and
    changes the channel count.  ``make_block(cin, cout, k, s, p, d, t)`` builds one block."""
    levels = []
    for cin, cout, k, s, p, d, t, n in rows:
        levels.append(nn.Sequential(*[make_block(cin if i == 0 else cout, cout, k, s if i == 0 else 1, p, d, t) for i in range(n)]))
    return levels


class ImageFill(_UNetBase):
    """MobileNetV2-style partial-conv U-Net (models/image_inpainting.py:9-86): 7x7 stem, three encoder levels of
    inverted residuals (t = 4), three dilated levels (2 / 4 / 8), three decoder levels (t = 2) on up-sampled +
    skip features, 3x3 head over the concatenation with the raw input."""

    def __init__(self):
        super().__init__()
        act = self.act_fn = nn.LeakyReLU(0.3)
        self.double_upscale = DoubleUpSample(scale_factor=2, mode="nearest")

        def pir(**flags):
            return lambda cin, cout, k, s, p, d, t: PartialInvertedResidual(cin, cout, k, s, p, d, t, bias=False, BN=True,
                                                                            activation=act, same_holes=True, **flags)
        cfg = IMAGEFILL
        self.encoder = nn.Sequential(partial_convolution_block(*cfg["stem"], bias=True, BN=False, activation=act),
                                     *build_levels(cfg["encoder"], pir(use_1_conv=True)))
        self.dilated_layers = nn.Sequential(*build_levels(cfg["dilated"], pir(no_holes_1_conv=True)))
        self.decoder = nn.Sequential(*build_levels(cfg["decoder"], pir(no_holes_1_conv=True)),
                                     partial_convolution_block(*cfg["head"], bias=True, BN=False, activation=False))

    def forward(self, args):
        x, mask = args                      # mask: 1 = known pixel, 0 = hole
        x, mp, fx, fm = self._encode(to_nhwc(x), as_parts(mask))
        x, mp = run_nhwc(self.dilated_layers, x, mp)
        return to_nchw(self._decode(x, mp, fx, fm))


class ImageFillOrigin(_UNetBase):
    """The partial-convolution paper's U-Net (models/image_inpainting.py:110-191): eight stride-2 encoder convs
    (ReLU, same_holes), seven decoder convs (LeakyReLU 0.2, per-channel hole bookkeeping) and a 3x3 head."""

    def __init__(self):
        super().__init__()
        self.double_upscale = DoubleUpSample(scale_factor=2, mode="nearest")

        def conv(act, same_holes):
            return lambda cin, cout, k, s, p, d, t: partial_convolution_block(cin, cout, k, s, p, d, groups=1, BN=True, activation=act,
                                                                              bias=False, no_holes_1_conv=False, same_holes=same_holes)
        cfg = ORIGIN
        self.encoder = nn.Sequential(partial_convolution_block(*cfg["stem"], bias=True, BN=False, activation=nn.ReLU(), same_holes=True),
                                     *build_levels(cfg["encoder"], conv(nn.ReLU(), True)))
        self.decoder = nn.Sequential(*build_levels(cfg["decoder"], conv(nn.LeakyReLU(0.2), False)),
                                     partial_convolution_block(*cfg["head"], bias=True, BN=False, activation=False, same_holes=False))

    def forward(self, args):
        x, mask = args
        x, mp, fx, fm = self._encode(to_nhwc(x), as_parts(mask))
        return to_nchw(self._decode(x, mp, fx, fm))


class DoublePartialResidual(BaseModule):
    """Two partial-conv blocks, ``conv2(conv1(x)) + conv1(x)`` (models/image_inpainting.py:194-216).  Like the
    reference, the ``padding`` / ``dilation`` arguments are ignored: both come from ``dilation_rate``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, expansion=1, BN=True, activation=True, bias=False,
                 use_1_conv=False, no_holes_1_conv=False, same_holes=False,
                 dilation_rate=(1, 1), *args, **kwargs):
        super().__init__()
        flags = dict(BN=BN, activation=activation, bias=bias, use_1_conv=use_1_conv, no_holes_1_conv=no_holes_1_conv,
                     same_holes=same_holes)
        r1, r2 = dilation_rate
        self.conv1 = partial_convolution_block(in_channels, out_channels, kernel_size, stride, padding=r1, dilation=r1, **flags)
        self.conv2 = partial_convolution_block(out_channels, out_channels, kernel_size, 1, padding=r2, dilation=r2, **flags)

    def forward_nhwc(self, x, mp):
        from .partial_convolution import PartialActivatedBN
        x1, m1 = run_nhwc(self.conv1, x, mp)
        if len(self.conv2) == 2 and isinstance(self.conv2[1], PartialActivatedBN):
            h, m2 = self.conv2[0].forward_nhwc(x1, m1)
            return self.conv2[1].forward_nhwc(h, m2, residual=x1)                # the add rides in the BatchNorm apply kernel
        x2, m2 = run_nhwc(self.conv2, x1, m1)
        return ops.add_act(x2, x1), m2

    def forward(self, args):
        x, mask = args
        keep_parts = isinstance(mask, MaskParts)
        y, mp = self.forward_nhwc(to_nhwc(x), as_parts(mask))
        return to_nchw(y), (mp if keep_parts else mp.as_tensor())


class ImageFillOriginV2(_UNetBase):
    """ImageFillOrigin with a DoublePartialResidual per level (encoder rates (1, 2), decoder (2, 1)), a 5x5 stem with
    BatchNorm and a ReLU after the head (models/image_inpainting.py:219-290)."""

    def __init__(self):
        super().__init__()
        self.double_upscale = DoubleUpSample(scale_factor=2, mode="nearest")
        act = nn.LeakyReLU(0.2)

        def dpr(same_holes, rates):
            return lambda cin, cout, k, s, p, d, t: DoublePartialResidual(cin, cout, k, s, p, d, BN=True, activation=act, bias=False,
                                                                          same_holes=same_holes, dilation_rate=rates)
        cfg = ORIGIN_V2
        self.encoder = nn.Sequential(partial_convolution_block(*cfg["stem"], bias=False, BN=True, activation=act, same_holes=True),
                                     *build_levels(cfg["encoder"], dpr(True, (1, 2))))
        self.decoder = nn.Sequential(*build_levels(cfg["decoder"], dpr(False, (2, 1))),
                                     partial_convolution_block(*cfg["head"], bias=True, BN=False, activation=nn.ReLU(), same_holes=False))

    def forward(self, args):
        x, mask = args
        x, mp, fx, fm = self._encode(to_nhwc(x), as_parts(mask))
        return to_nchw(self._decode(x, mp, fx, fm))

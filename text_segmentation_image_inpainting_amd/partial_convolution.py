"""Host-side mirror of the reference's hard-gated partial convolution family
(models/partial_convolution.py:20-231): same class names, constructor signatures,
``forward((x, mask)) -> (out, new_mask)`` surface and ``state_dict`` keys
(``feature_conv.weight/bias``, frozen all-ones ``mask_conv.weight``, ``bn_act.0.*``).

The nn.Conv2d / nn.BatchNorm2d members only hold parameters (so keys, shapes and default
initialisation match the reference); they are never called.  ``forward`` dispatches to the
HIP kernels through ops.py: the all-ones mask convolution is replaced by the K1 box count on
mask planes, x*mask / division / hole zeroing are fused into the conv kernels.
Each module also has ``forward_nhwc(x_nhwc, MaskParts)``, the zero-copy internal protocol
used by the mirrored networks.
"""
import torch
from torch import nn

from . import ops
from .BaseModels import BaseModule, act_code, bn_state, to_nchw, to_nhwc
from .masks import MaskParts, as_parts

# A/B switches (module attributes, see ops.py): K6b BatchNorm folding -- "1" (default) statistics in the conv epilogue + apply on
# load, "stats" statistics only, "0" the separate-kernel path
FUSE_BN = "1"
# K7b (1x1 convolutions over a decoder concatenation run their low half at low resolution): False = materialise the
# concatenation (K7) and run one product over it, as the reference does
FUSE_UPCAT = True

inplace_batch_norm = False  # reference: optional un-vendored InPlaceABN (:12-17); never available


def _public_forward(module, args, **kw):
    x, mask = args
    keep_parts = isinstance(mask, MaskParts)
    y, mp = module.forward_nhwc(to_nhwc(x), as_parts(mask), **kw)
    return to_nchw(y), (mp if keep_parts else mp.as_tensor())


class PartialConv(BaseModule):
    # mask is binary, 0 is holes; 1 is not            (models/partial_convolution.py:20-80)
    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias=True, same_holes=False):
        super().__init__()
        self.feature_conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride,
                                      padding, dilation, groups, bias)
        nn.init.kaiming_normal_(self.feature_conv.weight)                       # :35
        self.same_holes = same_holes
        mask_in = 1 if same_holes else in_channels                              # :38-40
        mask_out = 1 if same_holes else out_channels
        mask_groups = 1 if same_holes else groups
        self.mask_conv = nn.Conv2d(mask_in, mask_out, kernel_size, stride,
                                   padding, dilation, mask_groups, bias=False)
        torch.nn.init.constant_(self.mask_conv.weight, 1.0)                     # :44-47
        for param in self.mask_conv.parameters():
            param.requires_grad = False

    fill_holes = True

    def _geom(self):
        fc = self.feature_conv
        return ops.make_geom(fc.kernel_size, fc.stride, fc.padding, fc.dilation)

    def forward_nhwc(self, x, mp, want_stats=False):
        """``x``: NHWC tensor or ops.LazyBN (a BatchNorm output still to be applied on load, K6b).  With
        ``want_stats`` returns (y, mask, stat_part or None) for the BatchNorm that follows."""
        fc = self.feature_conv
        w, b = fc.weight, fc.bias
        cin, cout, groups = fc.in_channels, fc.out_channels, fc.groups
        g = self._geom()
        if x.shape[-1] != cin or mp.channels != cin:
            raise RuntimeError(f"PartialConv expects {cin} input channels, got x:{x.shape[-1]} mask:{mp.channels}")
        # K1: valid-input count -> denom / new mask / reciprocal planes          (:57-66,74-75)
        if self.same_holes:
            p0, a0, p1, a1, post = mp.first_channel_plane(), 1.0, None, 0.0, float(cin)   # :59-61
        elif groups == 1:
            (p0, a0, p1, a1), post = mp.count_operands(), 1.0                            # :63
        else:
            raise NotImplementedError("grouped PartialConv without same_holes (per-group mask counts, "
                                      "models/partial_convolution.py:63) is not used by any reference "
                                      "network and has no HIP kernel")
        denom, new_mask, inv = ops.mask_update(p0, a0, p1, a1, g, post, self.fill_holes)
        keep = new_mask if self.fill_holes else None
        pointwise = tuple(g) == (1, 1, 1, 1, 0, 0, 1, 1)
        part = None
        real = lambda t: t.materialize() if isinstance(t, ops.LazyBN) else t
        if isinstance(x, ops.VirtualCat):
            if (groups == 1 and mp.fusable and not want_stats and len(mp.parts) == 2 and mp.parts[0].channels == x.low.shape[3]
                    and ops.head_cat_ok(x, cout, g)):
                # K4c: the decoder's up-sample + concat is consumed by the head without being written
                r0, _, r1 = mp.row_scale()
                y = ops.pconv_head_cat(x, w, b, r0, r1, denom, keep, inv)
                return y, MaskParts.from_plane(new_mask, cout)
            if (groups == 1 and pointwise and mp.fusable and len(mp.parts) == 2 and mp.parts[0].channels == x.low.shape[3]
                    and FUSE_UPCAT and ops.pointwise_up_ok(x, cout)):
                # K7b: conv1x1(cat(up2(low), skip) * mask) = up2(conv1x1(low * mask_low)) + conv1x1(skip * mask_skip): the low
                # half at low resolution (a quarter of its multiply-adds), the concatenation never written
                _, cl, r1 = mp.row_scale()                 # r1: the skip part's plane (None: premultiplied / all ones)
                z = ops.pconv_pointwise(x.low, w[:, :cl], None, x.low_plane, cl, None)
                y = ops.pconv_pointwise(x.skip, w[:, cl:], b, r1, cin - cl, None, denom, keep, inv, want_stats=want_stats, up_add=z)
                if want_stats:
                    y, part = y
                new_mp = MaskParts.from_plane(new_mask, cout)
                return (y, new_mp, part) if want_stats else (y, new_mp)
            x = x.materialize()
        if groups == 1:
            if mp.fusable:
                r0, split, r1 = mp.row_scale()
                if pointwise:
                    y = ops.pconv_pointwise(x, w, b, r0, split, r1, denom, keep, inv, want_stats=want_stats)
                    if want_stats:
                        y, part = y
                else:
                    y = ops.pconv_dense(real(x), w, b, None, r0, split, r1, denom, keep, inv, g, want_stats=want_stats)
                    if want_stats:
                        y, part = y
            else:
                mfull = mp.full_nhwc()
                if pointwise:
                    y = ops.pconv_pointwise(ops.mul_mask(real(x), mfull), w, b, None, 0, None, denom, keep, inv,
                                            want_stats=want_stats)
                    if want_stats:
                        y, part = y
                else:
                    y = ops.pconv_dense(real(x), w, b, mfull, None, 0, None, denom, keep, inv, g, want_stats=want_stats)
                    if want_stats:
                        y, part = y
        elif groups == cin == cout:
            if mp.fusable and len(mp.parts) == 1:
                y = ops.pconv_depthwise(x, w, b, mp.parts[0].plane, denom, keep, inv, g, want_stats=want_stats)
            else:
                y = ops.pconv_depthwise(ops.mul_mask(real(x), mp.full_nhwc()), w, b, None, denom, keep, inv, g,
                                        want_stats=want_stats)
            if want_stats:
                y, part = y
        else:
            raise NotImplementedError(f"PartialConv groups={groups} (neither 1 nor depth-wise) has no HIP kernel")
        new_mp = MaskParts.from_plane(new_mask, cout)                                    # :74-77
        return (y, new_mp, part) if want_stats else (y, new_mp)

    def forward(self, args):
        return _public_forward(self, args)


class PartialConv1x1(BaseModule):
    """Plain 1x1 conv of x (no x*mask); mask = channel 0 expanded (models/partial_convolution.py:83-105)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1,
                 padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        assert kernel_size == 1 and stride == 1 and padding == 0                 # :96
        self.feature_conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride,
                                      padding, dilation, groups, bias)
        nn.init.kaiming_normal_(self.feature_conv.weight)

    def forward_nhwc(self, x, mp, want_stats=False):
        fc = self.feature_conv
        if fc.groups != 1:
            raise NotImplementedError("grouped PartialConv1x1 has no HIP kernel")
        y = ops.pconv_pointwise(x, fc.weight, fc.bias, want_stats=want_stats)
        new_mp = MaskParts.from_plane(mp.first_channel_plane(), fc.out_channels)   # :104
        if want_stats:
            return y[0], new_mp, y[1]
        return y, new_mp

    def forward(self, args):
        return _public_forward(self, args)


class PartialConvNoHoles(PartialConv):
    """No hole handling (0/0 -> NaN), new mask all ones (models/partial_convolution.py:108-137)."""

    fill_holes = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        assert self.feature_conv.groups == 1                                     # :119


class PartialActivatedBN(BaseModule):
    """(BatchNorm2d -> activation)(x), mask passthrough (models/partial_convolution.py:183-201)."""

    def __init__(self, channel, act_fn):
        super().__init__()
        if act_fn:
            self.bn_act = nn.Sequential(nn.BatchNorm2d(channel), act_fn)         # :195
        else:
            self.bn_act = nn.Sequential(nn.BatchNorm2d(channel))                 # :197

    def _cfg(self):
        bn = self.bn_act[0]
        act, slope = act_code(self.bn_act[1] if len(self.bn_act) > 1 else None)
        training, momentum, rmean, rvar = bn_state(bn)
        return bn, act, slope, training, momentum, rmean, rvar

    def forward_nhwc(self, x, mp, residual=None):
        bn, act, slope, training, momentum, rmean, rvar = self._cfg()
        if isinstance(x, ops.LazyBN):
            x = x.materialize()
        y = ops.bn_act(x, bn.weight, bn.bias, rmean, rvar, training, momentum, bn.eps, act, slope, residual)
        return y, mp

    def forward_lazy(self, y, mp, part=None):
        """K6b: statistics from the partials the producing conv left behind (``part``); the normalised activation
        stays virtual (ops.LazyBN) until a consumer loads it or ``materialize()`` writes it."""
        bn, act, slope, training, momentum, rmean, rvar = self._cfg()
        if bn.weight is None:
            raise NotImplementedError("BatchNorm2d(affine=False) is not used by the reference networks")
        lazy = ops.bn_lazy(y, bn.weight, bn.bias, rmean, rvar, training, momentum, bn.eps, act, slope,
                           part if training else None)
        return lazy, mp

    def forward(self, args):
        return _public_forward(self, args)


class PartialActivation(BaseModule):
    def __init__(self, activation):                                              # :204-211
        super().__init__()
        self.act_fn = activation

    def forward_nhwc(self, x, mp):
        act, slope = act_code(self.act_fn)
        return ops.activation(x, act, slope), mp

    def forward(self, args):
        return _public_forward(self, args)


class DoubleUpSample(nn.Module):
    """Nearest x2 on x and on the mask (models/partial_convolution.py:224-231)."""

    def __init__(self, scale_factor, mode="nearest"):
        super().__init__()
        if scale_factor != 2 or mode != "nearest":
            raise NotImplementedError("only the nearest x2 up-sampling the reference networks use has a HIP kernel")
        self.upsample = nn.Upsample(scale_factor=scale_factor, mode=mode)  # attribute kept for parity; never called

    def forward_nhwc(self, x, mp):
        if not (mp.fusable and len(mp.parts) == 1 and mp.parts[0].planar):
            # general per-channel mask: up-sample it like a feature map
            up = ops.upsample2x(mp.full_nhwc())
            from .masks import Part
            return ops.upsample2x(x), MaskParts([Part(mp.channels, full=up)])
        return ops.upsample2x(x), mp.upsample2x()

    def forward(self, args):
        return _public_forward(self, args)


def run_block(block, x, mp, allow_lazy=False, residual=None):
    """Run a ``partial_convolution_block`` Sequential with its BatchNorm folded into the neighbouring convs (K6b):
    the conv emits the BatchNorm partial sums, the BatchNorm becomes a LazyBN that the next conv applies on load.
    ``allow_lazy``: hand the LazyBN to the caller instead of writing the activation; ``residual`` is added when the
    trailing BatchNorm is written (MobileNetV2.py:186-187)."""
    mods = list(block)
    conv_types = (PartialConv, PartialConv1x1)
    if FUSE_BN != "0" and len(mods) == 2 and isinstance(mods[0], conv_types) and isinstance(mods[1], PartialActivatedBN):
        conv, bn = mods
        bn0 = bn.bn_act[0]
        if bn0.training or bn0.running_mean is None:
            y, m, part = conv.forward_nhwc(x, mp, want_stats=True)
        else:                                        # eval: running statistics, nothing to collect
            (y, m), part = conv.forward_nhwc(x, mp), None
        lazy, m = bn.forward_lazy(y, m, part)
        if allow_lazy and residual is None and FUSE_BN == "1" and ops.load_time_act(lazy.act, lazy.slope):
            return lazy, m
        return lazy.materialize(residual), m
    for mod in mods:
        if isinstance(x, ops.LazyBN) and not isinstance(mod, conv_types):
            x = x.materialize()
        x, mp = mod.forward_nhwc(x, mp)
    if isinstance(x, ops.LazyBN):
        x = x.materialize()
    if residual is not None:
        x = ops.add_act(x, residual)
    return x, mp


def partial_convolution_block(in_channels, out_channels, kernel_size, stride=1, padding=0,
                              dilation=1, groups=1, bias=False, BN=True, activation=True,
                              use_1_conv=False, no_holes_1_conv=False, same_holes=False):
    """models/partial_convolution.py:163-180 (note: same_holes only reaches PartialConv, :173-174)."""
    if use_1_conv:
        m = [PartialConv1x1(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)]
    elif no_holes_1_conv:
        m = [PartialConvNoHoles(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)]
    else:
        m = [PartialConv(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, same_holes)]
    if BN:
        m += [PartialActivatedBN(out_channels, activation)]
    if not BN and activation:
        m += [PartialActivation(activation)]
    return nn.Sequential(*m)

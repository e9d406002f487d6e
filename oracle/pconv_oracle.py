"""Functional CPU oracle of the partial-convolution inpainting path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Stock PyTorch ops on
CPU tensors, written from SURVEY.md section 8(a) and the reference's observable
behaviour; every function cites the reference lines it restates
(paths relative to the reference repo root).  Parameters come from a
``state_dict`` that uses the reference's key names, so the same dict can be
loaded into the product modules and fed to the oracle.

dtype follows the inputs: run it in float64 for a noise-floor reference.
"""
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ---------------------------------------------------------------------------
# a1..a3  partial convolution family  (models/partial_convolution.py)
# ---------------------------------------------------------------------------
def partial_conv(x: Tensor, mask: Tensor, weight: Tensor, bias: Optional[Tensor],
                 stride=1, padding=0, dilation=1, groups=1, same_holes=False):
    """``PartialConv.forward`` (models/partial_convolution.py:49-80).

    out = (conv(x*m)+b - b)/cnt + b, zero where the window saw no valid pixel;
    cnt = all-ones conv of the mask (:57-64); same_holes counts channel 0 only
    and multiplies by Cin (:58-61) -- also for depth-wise convs (quirk F6).
    """
    cin = x.shape[1]
    cout, cin_g, kh, kw = weight.shape
    y = F.conv2d(x * mask, weight, bias, stride, padding, dilation, groups)
    if bias is not None:
        b = bias.view(1, -1, 1, 1).expand_as(y)
    else:
        b = torch.zeros_like(y)
    with torch.no_grad():
        if same_holes:
            ones = torch.ones(1, 1, kh, kw, dtype=x.dtype)
            cnt = F.conv2d(mask[:, :1], ones, None, stride, padding, dilation, 1)
            hole = cnt == 0
            cnt = cnt * cin
        else:
            ones = torch.ones(cout, cin_g, kh, kw, dtype=x.dtype)
            cnt = F.conv2d(mask, ones, None, stride, padding, dilation, groups)
            hole = cnt == 0
        denom = cnt.masked_fill(hole, 1.0)
    out = ((y - b) / denom + b).masked_fill(hole, 0.0)
    new_mask = torch.ones_like(cnt).masked_fill(hole, 0.0)
    if same_holes:
        new_mask = new_mask.expand_as(out)
    return out, new_mask


def partial_conv1x1(x: Tensor, mask: Tensor, weight: Tensor, bias: Optional[Tensor]):
    """``PartialConv1x1.forward`` (models/partial_convolution.py:101-105):
    plain 1x1 conv of x (no x*m), mask = channel 0 expanded."""
    out = F.conv2d(x, weight, bias)
    return out, mask[:, :1].expand_as(out)


def partial_conv_noholes(x: Tensor, mask: Tensor, weight: Tensor, bias: Optional[Tensor],
                         stride=1, padding=0, dilation=1):
    """``PartialConvNoHoles.forward`` (models/partial_convolution.py:121-137):
    no hole handling (0/0 -> NaN), new mask is all ones."""
    cout, cin_g, kh, kw = weight.shape
    y = F.conv2d(x * mask, weight, bias, stride, padding, dilation, 1)
    b = bias.view(1, -1, 1, 1).expand_as(y) if bias is not None else torch.zeros_like(y)
    with torch.no_grad():
        ones = torch.ones(cout, cin_g, kh, kw, dtype=x.dtype)
        cnt = F.conv2d(mask, ones, None, stride, padding, dilation, 1)
    out = (y - b) / cnt + b
    return out, torch.ones_like(out)


# ---------------------------------------------------------------------------
# a5  BN + activation, a6 up-sampling  (models/partial_convolution.py:183-231)
# ---------------------------------------------------------------------------
def leaky(slope: float) -> Callable[[Tensor], Tensor]:
    return lambda t: F.leaky_relu(t, slope)


def bn(sd: Dict[str, Tensor], prefix: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1); running stats in ``sd``
    are updated in place in training mode, like the module's buffers."""
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    converted = rm.dtype != x.dtype  # float64 noise-floor runs
    if converted:
        rm, rv = rm.to(x.dtype), rv.to(x.dtype)
    out = F.batch_norm(x, rm, rv, sd[prefix + "weight"].to(x.dtype), sd[prefix + "bias"].to(x.dtype),
                       training, 0.1, 1e-5)
    if training:
        if converted:
            sd[prefix + "running_mean"].copy_(rm)
            sd[prefix + "running_var"].copy_(rv)
        if prefix + "num_batches_tracked" in sd:
            sd[prefix + "num_batches_tracked"] += 1
    return out


def pconv_block(sd, prefix, x, mask, stride=1, padding=0, dilation=1, groups=1,
                BN=True, act=None, use_1_conv=False, no_holes_1_conv=False,
                same_holes=False, training=True):
    """``partial_convolution_block`` (models/partial_convolution.py:163-180): module 0 is
    the conv flavour, module 1 is ``PartialActivatedBN`` (``bn_act.0`` = BatchNorm2d,
    :193-197) when BN else ``PartialActivation`` when an activation is given."""
    w = sd[prefix + "0.feature_conv.weight"].to(x.dtype)
    b = sd.get(prefix + "0.feature_conv.bias")
    b = b.to(x.dtype) if b is not None else None
    if use_1_conv:
        x, mask = partial_conv1x1(x, mask, w, b)
    elif no_holes_1_conv:
        x, mask = partial_conv_noholes(x, mask, w, b, stride, padding, dilation)
    else:
        x, mask = partial_conv(x, mask, w, b, stride, padding, dilation, groups, same_holes)
    if BN:
        x = bn(sd, prefix + "1.bn_act.0.", x, training)
        if act:
            x = act(x)
    elif act:
        x = act(x)
    return x, mask


def double_upsample(x, mask):
    """``DoubleUpSample`` nearest x2 on both (models/partial_convolution.py:229-231)."""
    return (F.interpolate(x, scale_factor=2, mode="nearest"),
            F.interpolate(mask, scale_factor=2, mode="nearest"))


# ---------------------------------------------------------------------------
# a7  PartialInvertedResidual  (models/MobileNetV2.py:152-190)
# ---------------------------------------------------------------------------
def partial_inverted_residual(sd, prefix, x, mask, in_c, out_c, k, stride, padding, dilation,
                              expansion, act, use_1_conv, no_holes_1_conv, same_holes, training):
    mid = int(in_c * expansion)
    p = prefix + "conv."
    h, m = pconv_block(sd, p + "0.", x, mask, 1, 0, 1, 1, True, act,
                       use_1_conv, no_holes_1_conv, False, training)       # :170-172
    h, m = pconv_block(sd, p + "1.", h, m, stride, padding, dilation, mid, True, act,
                       False, False, same_holes, training)                 # :174-176
    h, m = pconv_block(sd, p + "2.", h, m, 1, 0, 1, 1, True, None,
                       use_1_conv, no_holes_1_conv, False, training)       # :178-180
    if stride == 1 and in_c == out_c:                                      # :158,186-187
        h = x + h
    return h, m


# ---------------------------------------------------------------------------
# a8  ImageFill  (models/image_inpainting.py:9-86)
# ---------------------------------------------------------------------------
_IF_ENC = [(64, 128, 3, 2, 1, 1, 4, 2), (128, 256, 3, 2, 1, 1, 4, 2), (256, 256, 3, 2, 1, 1, 4, 2)]
_IF_DIL = [(256, 256, 3, 1, 2, 2, 4, 2), (256, 256, 3, 1, 4, 4, 4, 2), (256, 256, 3, 1, 8, 8, 4, 2)]
_IF_DEC = [(512, 256, 3, 1, 1, 1, 2, 1), (384, 128, 3, 1, 1, 1, 2, 1), (192, 32, 3, 1, 1, 1, 2, 1)]


def _pir_stage(sd, prefix, x, mask, setting, act, use_1, no_holes, same_holes, training):
    in_c, out_c, k, s, p, d, t, n = setting
    for i in range(n):                                                     # make_layers :46-65
        x, mask = partial_inverted_residual(sd, f"{prefix}{i}.", x, mask, in_c, out_c, k,
                                            s if i == 0 else 1, p, d, t, act,
                                            use_1, no_holes, same_holes, training)
        in_c = out_c
    return x, mask


def image_fill(sd, x, mask, training=True, taps: Optional[dict] = None):
    act = leaky(0.3)                                                       # :12
    fx, fm = [x], [mask]
    # encoder[0]: 7x7 s2 stem, bias, no BN, non-same-holes                   :23
    x, mask = pconv_block(sd, "encoder.0.", x, mask, 2, 3, 1, 1, False, act, training=training)
    fx.append(x); fm.append(mask)
    for li, st in enumerate(_IF_ENC):                                      # :24
        x, mask = _pir_stage(sd, f"encoder.{li + 1}.", x, mask, st, act, True, False, True, training)
        fx.append(x); fm.append(mask)
    if taps is not None:
        taps["enc_masks"] = [m[:, :1].clone() for m in fm[1:]]
        taps["enc_out"] = x.clone()
    fx, fm = fx[:-1], fm[:-1]                                              # :77-78
    for li, st in enumerate(_IF_DIL):                                      # :33,79
        x, mask = _pir_stage(sd, f"dilated_layers.{li}.", x, mask, st, act, False, True, True, training)
    if taps is not None:
        taps["dil_out"] = x.clone()
    for li, st in enumerate(_IF_DEC):                                      # :81-85
        x, mask = double_upsample(x, mask)
        x = torch.cat([x, fx.pop(-1)], 1)
        mask = torch.cat([mask, fm.pop(-1)], 1)
        x, mask = _pir_stage(sd, f"decoder.{li}.", x, mask, st, act, False, True, True, training)
    x, mask = double_upsample(x, mask)
    x = torch.cat([x, fx.pop(-1)], 1)
    mask = torch.cat([mask, fm.pop(-1)], 1)
    x, mask = pconv_block(sd, "decoder.3.", x, mask, 1, 1, 1, 1, False, None, training=training)  # :44
    return x


# ---------------------------------------------------------------------------
# a9  ImageFillOrigin  (models/image_inpainting.py:110-191)
# ---------------------------------------------------------------------------
_IFO_ENC = [(64, 128, 5, 2, 2), (128, 256, 5, 2, 2), (256, 512, 3, 2, 1), (512, 512, 3, 2, 1),
            (512, 512, 3, 2, 1), (512, 512, 3, 2, 1), (512, 512, 3, 2, 1)]
_IFO_DEC = [(1024, 512), (1024, 512), (1024, 512), (1024, 512), (768, 256), (384, 128), (192, 64)]


def image_fill_origin(sd, x, mask, training=True):
    relu = F.relu
    fx, fm = [x], [mask]
    x, mask = pconv_block(sd, "encoder.0.", x, mask, 2, 3, 1, 1, False, relu, same_holes=True,
                          training=training)                               # :132
    fx.append(x); fm.append(mask)
    for li, (ic, oc, k, s, p) in enumerate(_IFO_ENC):                      # make_layer_v2 :155-162
        x, mask = pconv_block(sd, f"encoder.{li + 1}.0.", x, mask, s, p, 1, 1, True, relu,
                              same_holes=True, training=training)
        fx.append(x); fm.append(mask)
    fx, fm = fx[:-1], fm[:-1]
    for li, (ic, oc) in enumerate(_IFO_DEC):                               # :151-153
        x, mask = double_upsample(x, mask)
        x = torch.cat([x, fx.pop(-1)], 1)
        mask = torch.cat([mask, fm.pop(-1)], 1)
        x, mask = pconv_block(sd, f"decoder.{li}.0.", x, mask, 1, 1, 1, 1, True, leaky(0.2),
                              same_holes=False, training=training)
    x, mask = double_upsample(x, mask)
    x = torch.cat([x, fx.pop(-1)], 1)
    mask = torch.cat([mask, fm.pop(-1)], 1)
    x, mask = pconv_block(sd, "decoder.7.", x, mask, 1, 1, 1, 1, False, None, training=training)
    return x


# ---------------------------------------------------------------------------
# a10  DoublePartialResidual / ImageFillOriginV2  (models/image_inpainting.py:194-290)
# ---------------------------------------------------------------------------
_IFV2_ENC = [(64, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512)]
_IFV2_DEC = [(1024, 512), (1024, 512), (768, 256), (512, 256), (512, 256), (384, 128), (192, 64)]


def double_partial_residual(sd, prefix, x, mask, stride, rates, act, same_holes, training):
    """:201-216 -- padding/dilation arguments are ignored, only ``dilation_rate`` is used."""
    x1, m1 = pconv_block(sd, prefix + "conv1.", x, mask, stride, rates[0], rates[0], 1, True, act,
                         same_holes=same_holes, training=training)
    x2, m2 = pconv_block(sd, prefix + "conv2.", x1, m1, 1, rates[1], rates[1], 1, True, act,
                         same_holes=same_holes, training=training)
    return x2 + x1, m2


def image_fill_origin_v2(sd, x, mask, training=True):
    act = leaky(0.2)
    fx, fm = [x], [mask]
    x, mask = pconv_block(sd, "encoder.0.", x, mask, 2, 2, 1, 1, True, act, same_holes=True,
                          training=training)                               # :241-242
    fx.append(x); fm.append(mask)
    for li, _ in enumerate(_IFV2_ENC):
        x, mask = double_partial_residual(sd, f"encoder.{li + 1}.0.", x, mask, 2, (1, 2), act, True, training)
        fx.append(x); fm.append(mask)
    fx, fm = fx[:-1], fm[:-1]
    for li, _ in enumerate(_IFV2_DEC):
        x, mask = double_upsample(x, mask)
        x = torch.cat([x, fx.pop(-1)], 1)
        mask = torch.cat([mask, fm.pop(-1)], 1)
        x, mask = double_partial_residual(sd, f"decoder.{li}.0.", x, mask, 1, (2, 1), act, False, training)
    x, mask = double_upsample(x, mask)
    x = torch.cat([x, fx.pop(-1)], 1)
    mask = torch.cat([mask, fm.pop(-1)], 1)
    x, mask = pconv_block(sd, "decoder.7.", x, mask, 1, 1, 1, 1, False, F.relu, training=training)  # :259-260
    return x


MODELS = {
    "ImageFill": image_fill,
    "ImageFillOrigin": image_fill_origin,
    "ImageFillOriginV2": image_fill_origin_v2,
}


def l1_mean(out: Tensor, target: Tensor) -> Tensor:
    """Throughput-benchmark loss of SURVEY.md 8(d) cfg 2: mean |out - clean| (nn.L1Loss)."""
    return (out - target).abs().mean()

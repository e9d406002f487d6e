"""Functional CPU oracle of the text-segmentation path (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Stock PyTorch ops on CPU, keyed by the reference's ``state_dict`` names; every function cites the
reference lines it restates.  Covers SURVEY.md 8(a) rows a11-a19: ``Conv_block`` / ``DSConvBlock``
(models/BaseModels.py:91-127), ``InvertedResidual`` / ``DilatedMobileNetV2`` (models/MobileNetV2.py),
``SpatialChannelSqueezeExcitation`` / ``ASP`` / ``RFB`` (models/common.py), ``ResidualBlock`` /
``Xception`` (models/Xception.py), ``TextSegament`` / ``XceptionTextSegment``
(models/text_segmentation.py) and ``BinaryFocalLoss`` (loss.py:58-83).
"""
import torch
import torch.nn.functional as F

from .pconv_oracle import bn, leaky


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def conv_block(sd, p, i, x, stride=1, padding=0, dilation=1, groups=1, BN=False, act=None, training=True):
    """``Conv_block`` placed at index ``i`` of the Sequential with prefix ``p`` (models/BaseModels.py:91-102):
    module i = Conv2d, module i+1 = Sequential(BatchNorm2d[, act]) when BN, else the bare activation.
    Returns (y, next_index)."""
    w = sd[f"{p}{i}.weight"].to(x.dtype)
    b = sd.get(f"{p}{i}.bias")
    b = b.to(x.dtype) if b is not None else None
    x = F.conv2d(x, w, b, _pair(stride), _pair(padding), _pair(dilation), groups)
    nxt = i + 1
    if BN:
        x = bn(sd, f"{p}{i + 1}.0.", x, training)
        if act:
            x = act(x)
        nxt = i + 2
    elif act is not None:
        x = act(x)
        nxt = i + 2
    return x, nxt


def scse(sd, p, x, act):
    """``SpatialChannelSqueezeExcitation.forward`` (models/common.py:32-43)."""
    b, c, h, w = x.shape
    ch = x.mean(dim=(2, 3))
    ch = F.linear(ch, sd[p + "channel_excite.0.weight"].to(x.dtype), sd[p + "channel_excite.0.bias"].to(x.dtype))
    ch = act(ch)
    ch = F.linear(ch, sd[p + "channel_excite.2.weight"].to(x.dtype), sd[p + "channel_excite.2.bias"].to(x.dtype))
    cse = torch.sigmoid(ch).view(b, c, 1, 1)
    sse = torch.sigmoid(F.conv2d(x, sd[p + "spatial_excite.0.weight"].to(x.dtype)))
    return x * cse + x * sse


def inverted_residual(sd, p, x, in_c, out_c, stride, t, d, act, add_sece, training):
    """``InvertedResidual`` (models/MobileNetV2.py:114-149)."""
    mid = in_c * t
    q = p + "conv."
    h, i = conv_block(sd, q, 0, x, 1, 0, 1, 1, True, act, training)                      # :133-135
    h, i = conv_block(sd, q, i, h, stride, 1 + (d - 1), d, mid, True, act, training)     # :136-139
    h, i = conv_block(sd, q, i, h, 1, 0, 1, 1, True, None, training)                     # :141
    if add_sece:
        h = scse(sd, f"{q}{i}.", h, act)                                                # :142-143
    return x + h if (stride == 1 and in_c == out_c) else h                               # :128,146-149


def make_divisible(v, divisor=8):
    """models/MobileNetV2.py:94-104."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


DILATED_SETTING = [(1, 16, 1, 1, 1), (6, 24, 2, 2, 1), (6, 32, 3, 2, 1), (6, 64, 4, 1, 2),
                   (6, 96, 3, 1, 4), (6, 160, 3, 1, 8), (6, 320, 1, 1, 16)]   # models/MobileNetV2.py:206-214


def mobilenet_stage_channels(width_mult=2):
    chans = [make_divisible(32 * width_mult)]
    for t, c, n, s, d in DILATED_SETTING:
        chans.append(make_divisible(c * width_mult))
    return chans


def mobilenet_feature(sd, p, idx, x, width_mult, act, add_sece, training):
    """``features[idx]`` of (Dilated)MobileNetV2 (models/MobileNetV2.py:46-69)."""
    chans = mobilenet_stage_channels(width_mult)
    if idx == 0:
        y, _ = conv_block(sd, f"{p}0.", 0, x, 2, 1, 1, 1, True, act, training)           # first layer :50-52
        return y
    t, c, n, s, d = DILATED_SETTING[idx - 1]
    in_c, out_c = chans[idx - 1], chans[idx]
    for i in range(n):
        x = inverted_residual(sd, f"{p}{idx}.{i}.", x, in_c, out_c, s if i == 0 else 1, t, d, act, add_sece, training)
        in_c = out_c
    return x


def rfb(sd, p, x, out_c, act, training):
    """``RFB`` with add_sece=True (models/common.py:96-156)."""
    outs = []
    # branch 0: conv_kernel 1, no half_conv (:139-144)
    q = p + "rfb.0."
    h, i = conv_block(sd, q, 0, x, 1, 0, 1, 1, True, act, training)
    h, i = conv_block(sd, q, i, h, 1, 1, 1, out_c, True, act, training)
    outs.append(h)
    for bi, (k, rate) in enumerate(((3, 5), (5, 17), (7, 29))):                          # :102,116-121
        q = f"{p}rfb.{bi + 1}."
        h, i = conv_block(sd, q, 0, x, 1, 0, 1, 1, True, act, training)                  # 1x1 -> mid
        h, i = conv_block(sd, q, i, h, 1, (0, (k - 1) // 2), 1, 1, True, None, training)  # (1,k)
        h, i = conv_block(sd, q, i, h, 1, ((k - 1) // 2, 0), 1, 1, True, None, training)  # (k,1)
        h, i = conv_block(sd, q, i, h, 1, rate, rate, out_c, True, act, training)        # dw 3x3 dilated
        outs.append(h)
    pool = torch.cat(outs, 1)
    pool = F.conv2d(pool, sd[p + "rfb_linear_conv.0.weight"].to(x.dtype), sd[p + "rfb_linear_conv.0.bias"].to(x.dtype))
    pool = scse(sd, p + "rfb_linear_conv.1.", pool, act)                                 # :108-111
    resi, _ = conv_block(sd, p + "input_down_channel.", 0, x, 1, 0, 1, 1, True, act, training)   # :105-106
    return act(pool + resi)                                                              # :156


def text_segament(sd, x, training=True, width_mult=2):
    """``TextSegament.forward`` (models/text_segmentation.py:60-84)."""
    act = leaky(0.3)
    chans = mobilenet_stage_channels(width_mult)
    outs = []
    for idx in range(3):
        x = mobilenet_feature(sd, "encoder.features.", idx, x, width_mult, act, True, training)
        outs.append(x)
    outs[0] = F.avg_pool2d(outs[0], 3, 2, 1)
    outs[1] = F.avg_pool2d(outs[1], 3, 2, 1)
    layer_out = torch.cat(outs, 1)
    pooled = []
    for idx in range(3, 8):
        x = mobilenet_feature(sd, "encoder.features.", idx, x, width_mult, act, True, training)
        pooled.append(x)
    x = rfb(sd, "feature_pooling.", torch.cat(pooled, 1), 256, act, training)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    concat_c = sum(chans[:3])
    layer_out = inverted_residual(sd, "feature_4x_conv.", layer_out, concat_c, 128, 1, 1, 1, act, True, training)
    x = torch.cat([layer_out, x], 1)
    x = inverted_residual(sd, "smooth_feature_4x_conv.0.", x, 256 + 128, 128, 1, 1, 2, act, True, training)
    x = inverted_residual(sd, "smooth_feature_4x_conv.1.", x, 128, 128, 1, 1, 1, act, True, training)
    x = F.conv2d(x, sd["out_conv.0.weight"].to(x.dtype), sd["out_conv.0.bias"].to(x.dtype), 1, 1)
    return F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)


# ---- Xception (models/Xception.py) --------------------------------------------------------
def ds_conv(sd, p, x, k, stride, padding, dilation, act_dep, act_point, training):
    """``DSConvBlock`` (models/BaseModels.py:105-127), bias=False, BN=True."""
    c = x.shape[1]
    x, _ = conv_block(sd, p + "depth_wise_conv.", 0, x, stride, padding, dilation, c, True, act_dep, training)
    x, _ = conv_block(sd, p + "point_wise_conv.", 0, x, 1, 0, 1, 1, True, act_point, training)
    return x


def residual_block(sd, p, x, in_c, out_c, stride, padding, dilation, act, training):
    """``ResidualBlock`` (models/Xception.py:13-44), expand_channel_first=True."""
    h = ds_conv(sd, p + "conv.0.", x, 3, 1, padding, dilation, act, act, training)
    h = ds_conv(sd, p + "conv.1.", h, 3, 1, padding, dilation, act, act, training)
    h = ds_conv(sd, p + "conv.2.", h, 3, stride, padding, dilation, act, None, training)
    r = x
    if stride > 1 or in_c != out_c:
        r, _ = conv_block(sd, p + "residual_conv.", 0, x, stride, 0, 1, 1, True, None, training)
    return h + r


def xception(sd, p, x, act, training):
    """``Xception.forward`` (models/Xception.py:108-114)."""
    q = p + "entry_flow_1."
    x, i = conv_block(sd, q, 0, x, 2, 1, 1, 1, True, act, training)
    x, i = conv_block(sd, q, i, x, 1, 1, 1, 1, True, act, training)
    x = residual_block(sd, f"{q}{i}.", x, 64, 128, 2, 1, 1, act, training)
    x4 = x
    x = residual_block(sd, p + "entry_flow_2.0.", x, 128, 256, 2, 1, 1, act, training)
    x = residual_block(sd, p + "entry_flow_2.1.", x, 256, 512, 1, 2, 2, act, training)
    for i in range(8):
        r = 2 if i < 4 else 4
        x = residual_block(sd, f"{p}middle_flow.{i}.", x, 512, 512, 1, r, r, act, training)
    for i, r in enumerate((2, 2, 1, 1)):
        x = residual_block(sd, f"{p}exit_flow.{i}.", x, 512, 512, 1, r, r, act, training)
    return x, x4


def asp(sd, p, x, act, rates, training):
    """``ASP.forward`` (models/common.py:86-93)."""
    outs = []
    h, _ = conv_block(sd, p + "asp.0.", 0, x, 1, 1, 1, 1, True, act, training)
    outs.append(h)
    for bi, r in enumerate(rates):
        h = F.avg_pool2d(x, r, 1, (r - 1) // 2)
        h, _ = conv_block(sd, f"{p}asp.{bi + 1}.", 1, h, 1, r, r, 1, True, act, training)
        outs.append(h)
    out, _ = conv_block(sd, p + "out_conv.", 0, torch.cat(outs, 1), 1, 0, 1, 1, True, act, training)
    return out


def xception_text_segment(sd, x, training=True):
    """``XceptionTextSegment.forward`` (models/text_segmentation.py:104-114)."""
    act = leaky(0.3)
    x, x4 = xception(sd, "encoder.", x, act, training)
    x4, _ = conv_block(sd, "feature_4x_conv.", 0, x4, 1, 0, 1, 1, True, act, training)
    x = asp(sd, "feature_pooling.", x, act, (3, 5, 9), training)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    x = torch.cat([x, x4], 1)
    x, i = conv_block(sd, "out_conv.", 0, x, 1, 1, 1, 1, True, act, training)
    x = F.conv2d(x, sd[f"out_conv.{i}.weight"].to(x.dtype), sd[f"out_conv.{i}.bias"].to(x.dtype), 1, 1)
    return F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)


def binary_focal_loss(logits, target, gamma=0.0, background_weights=1.0, words_weights=2.0):
    """``BinaryFocalLoss.forward`` (loss.py:66-75)."""
    x = logits.reshape(-1, 1)
    t = target.reshape(-1, 1)
    w = torch.where(t > 0, torch.full_like(t, words_weights), torch.full_like(t, background_weights))
    pt = F.logsigmoid(-x * (t * 2 - 1))
    loss = F.binary_cross_entropy_with_logits(x, t, weight=w, reduction="none")
    return ((pt * gamma).exp() * loss).mean()


def gram_matrix(feat):
    """loss.py:294-300."""
    b, ch, h, w = feat.shape
    f = feat.reshape(b, ch, h * w)
    return torch.bmm(f, f.transpose(1, 2)) / (ch * h * w)


def total_variation_loss(image):
    """loss.py:303-307."""
    return (image[:, :, :, :-1] - image[:, :, :, 1:]).abs().mean() + (image[:, :, :-1, :] - image[:, :, 1:, :]).abs().mean()


def inpainting_loss(sd, raw_input, mask, output, origin, width_mult=1, feature_range=3, training=True):
    """``InpaintingLoss.forward`` (loss.py:195-225) with ``FeatureExtractor`` over the first stages of a
    MobileNetV2 (``sd`` keys: ``feature_encoder.layers.<i>....``; ReLU6, no scSE -- MobileNetV2 defaults)."""
    relu6 = F.relu6

    def feats(x):
        out = []
        for idx in range(feature_range):
            x = mobilenet_feature(sd, "feature_encoder.layers.", idx, x, width_mult, relu6, False, training)
            out.append(x)
        return out

    l1 = lambda a, b: (a - b).abs().mean()
    comp = mask * raw_input + (1 - mask) * output
    loss_validate = l1(mask * output, mask * origin)
    loss_hole = l1((1 - mask) * output, (1 - mask) * origin)
    loss_tv = total_variation_loss(comp)
    f_comp, f_out, f_org = feats(comp), feats(output), feats(origin)
    loss_perc = sum(l1(x, y) for x, y in zip(f_comp, f_org)) + sum(l1(x, y) for x, y in zip(f_out, f_org))
    loss_style = sum(l1(gram_matrix(x), gram_matrix(y)) for x, y in zip(f_out, f_org)) + \
        sum(l1(gram_matrix(x), gram_matrix(y)) for x, y in zip(f_comp, f_org))
    return 1.0 * loss_validate + 6.0 * loss_hole + 0.1 * loss_tv + 0.05 * loss_perc + 120 * loss_style


SEG_MODELS = {"TextSegament": text_segament, "XceptionTextSegment": xception_text_segment}

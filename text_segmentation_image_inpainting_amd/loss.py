"""Mirror of the on-path losses (loss.py): ``BinaryFocalLoss`` (:58-83), ``InpaintingLoss`` (:185-225) with its
``FeatureExtractor`` (:228-241), ``gram_matrix`` (:294-300) and ``total_variation_loss`` (:303-307)."""
from torch import nn

from . import ops
from .BaseModels import to_nhwc


class BinaryFocalLoss(nn.Module):
    # gamma 0 gives the best AP scores (loss.py:59)
    def __init__(self, gamma=0, background_weights=1, words_weights=2):
        super().__init__()
        self.gamma = gamma
        self.background_weights = background_weights
        self.words_weights = words_weights

    def forward(self, input, target):
        assert input.dim() == 4 and input.size(1) == 1      # flatten_images (:79)
        assert target.dim() == 4 and target.size(1) == 1
        # [N,1,H,W] -> [N*H*W, 1]: with one channel NCHW and NHWC orders coincide, so no permute is needed
        return ops.bce_focal(input.reshape(-1), target.reshape(-1), self.gamma, self.background_weights,
                             self.words_weights)


class FeatureExtractor(nn.Module):
    """First ``feature_range`` stages of a (mirrored) MobileNetV2 with frozen parameters (loss.py:228-241)."""

    def __init__(self, encoder, feature_range=3):
        super().__init__()
        self.layers = nn.Sequential(*[encoder.features[i] for i in range(feature_range)])
        for layer in self.layers:
            for param in layer.parameters():
                param.requires_grad = False

    def forward(self, x):
        out = []
        for layer in self.layers:
            x = layer(x)
            out.append(x)
        return out


def gram_matrix(feat):
    """[b,ch,h,w] features -> [b,ch,ch] Gram matrices / (ch*h*w)  (loss.py:294-300)."""
    return ops.gram_matrix(to_nhwc(feat))


def total_variation_loss(image):
    """mean |dx| + mean |dy| of an [N,C,H,W] image (loss.py:303-307)."""
    return ops.total_variation(to_nhwc(image))


class InpaintingLoss(nn.Module):
    # Image Inpainting for Irregular Holes Using Partial Convolutions, weights from the paper (loss.py:223-224)
    def __init__(self, feature_encoder, feature_range=3):
        super().__init__()
        self.feature_encoder = FeatureExtractor(feature_encoder, feature_range)

    @staticmethod
    def _l1(a, b):
        return ops.l1_mean(to_nhwc(a) if a.dim() == 4 else a, to_nhwc(b) if b.dim() == 4 else b)

    def forward(self, raw_input, mask, output, origin):
        raw, m, out, gt = to_nhwc(raw_input), to_nhwc(mask), to_nhwc(output), to_nhwc(origin)
        comp = ops.compose(raw, m, out)                                        # :196
        loss_pixel = ops.masked_l1(out, gt, m, 1.0, 6.0)                       # 1*valid + 6*hole (:199-200,223)
        loss_tv = ops.total_variation(comp)                                    # :203
        comp_nchw = comp.permute(0, 3, 1, 2)
        feature_comp = self.feature_encoder(comp_nchw)                         # :206-208
        feature_output = self.feature_encoder(output)
        feature_origin = self.feature_encoder(origin)
        loss_perceptual = sum(self._l1(x, y) for x, y in zip(feature_comp, feature_origin)) + \
            sum(self._l1(x, y) for x, y in zip(feature_output, feature_origin))     # :211-213
        loss_style = sum(self._l1(gram_matrix(x), gram_matrix(y)) for x, y in zip(feature_output, feature_origin)) + \
            sum(self._l1(gram_matrix(x), gram_matrix(y)) for x, y in zip(feature_comp, feature_origin))   # :216-220
        return loss_pixel + 0.1 * loss_tv + 0.05 * loss_perceptual + 120 * loss_style   # :223-224

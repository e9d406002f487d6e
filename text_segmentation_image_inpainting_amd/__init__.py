"""MI355X-native (gfx950) implementation of the partial-convolution inpainting hot path of
yu45020/Text_Segmentation_Image_Inpainting: the reference's nn.Module surface over hand-written
HIP kernels behind a C ABI (include/tsii_hip.h).  GPU only -- there is no CPU fallback."""
from .image_inpainting import DoublePartialResidual, ImageFill, ImageFillOrigin, ImageFillOriginV2  # noqa: F401
from .MobileNetV2 import PartialInvertedResidual  # noqa: F401
from .partial_convolution import (DoubleUpSample, PartialActivatedBN, PartialActivation, PartialConv,  # noqa: F401
                                  PartialConv1x1, PartialConvNoHoles, partial_convolution_block)
from .MobileNetV2 import DilatedMobileNetV2, InvertedResidual, MobileNetV2  # noqa: F401,E402
from .Xception import ResidualBlock, Xception  # noqa: F401,E402
from .common import ASP, RFB, SpatialChannelSqueezeExcitation  # noqa: F401,E402
from .text_segmentation import TextSegament, XceptionTextSegment  # noqa: F401,E402
from .loss import BinaryFocalLoss, FeatureExtractor, InpaintingLoss, gram_matrix, total_variation_loss  # noqa: F401,E402
from .recipes import InpaintingRecipe, SegmentationRecipe  # noqa: F401,E402
from .ops import activation_storage, set_activation_storage  # noqa: F401,E402

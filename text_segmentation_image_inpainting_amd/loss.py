"""Mirror of the on-path losses (loss.py): ``BinaryFocalLoss`` (:58-83).  The mean-L1 loss used by the
inpainting benchmark step is ``ops.l1_mean``; the full ``InpaintingLoss`` (:185-225) is a later scope row."""
from torch import nn

from . import ops


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

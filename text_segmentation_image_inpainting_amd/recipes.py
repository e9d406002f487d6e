"""The reference's training recipes as loops over the fused step (row n3 of SURVEY.md 8(f)).

The reference ships no training script; what it documents is
  * inpainting: ``InpaintingLoss`` through a frozen MobileNetV2 feature extractor (loss.py:185-241), SGD with Nesterov
    momentum, weight decay 1e-4, cyclical learning rate 1e-4 .. 1e-2 stepped once per batch (ReadME.md:184-200,
    models/utils/cls.py:74-157);
  * text segmentation: ``BinaryFocalLoss(0, 1, 2)``, the same optimizer (Xception checkpoint: weight decay 1e-3, cyclical
    1e-4 .. 4e-4, checkpoints/ReadME.md:4), in TWO STAGES: encoder frozen first, then every parameter re-trained
    (ReadME.md:160; ``MobileNetV2.freeze_params``, models/MobileNetV2.py:86-92).
Both are thin host-side loops: ``train_step.FlatSGDTrainer`` (flat parameter / gradient / momentum buffers, bucketed all-reduce,
one fused ``tsii_sgd_nesterov`` launch) does the work, ``utils.cls.CyclicLR`` drives its learning rate.  Frozen parameters
never enter the flat buffers; their BatchNorm layers keep updating running statistics exactly as in the reference, which only
clears ``requires_grad`` and leaves the modules in train mode.
"""
from torch import nn

from .loss import BinaryFocalLoss, InpaintingLoss
from .train_step import FlatSGDTrainer
from .utils.cls import CyclicLR


class _Recipe:
    def __init__(self, trainer, base_lr, max_lr, step_size, mode):
        self.trainer = trainer
        # the reference calls scheduler.batch_step() BEFORE train_batch (models/utils/cls.py:66-68): the constructor sets
        # iteration 0's rate, every step() advances first
        self.scheduler = CyclicLR(trainer, base_lr=base_lr, max_lr=max_lr, step_size=step_size, mode=mode)

    @property
    def lr(self):
        return self.trainer.lr


class InpaintingRecipe(_Recipe):
    """``step(corrupted, mask, clean)`` = forward, InpaintingLoss, backward, gradient all-reduce, SGD-Nesterov, one
    scheduler tick.  ``feature_encoder``: a ``MobileNetV2`` whose first ``feature_range`` stages give the perceptual /
    style features; it is frozen here (loss.py:232-234)."""

    def __init__(self, model, feature_encoder, feature_range=3, base_lr=1e-4, max_lr=1e-2, step_size=2000, mode="triangular",
                 momentum=0.9, weight_decay=1e-4, **trainer_kw):
        for p in feature_encoder.parameters():
            p.requires_grad_(False)
        self.criterion = InpaintingLoss(feature_encoder, feature_range)
        self._batch = None
        trainer = FlatSGDTrainer(model, lr=base_lr, momentum=momentum, weight_decay=weight_decay,
                                 loss_fn=lambda out, _unused: self.criterion(self._batch[0], self._batch[1], out, self._batch[2]), **trainer_kw)
        super().__init__(trainer, base_lr, max_lr, step_size, mode)

    def to(self, device):
        self.criterion.to(device)
        return self

    def step(self, corrupted, mask, clean):
        self.scheduler.batch_step()
        self._batch = (corrupted, mask, clean)
        return self.trainer.step(corrupted, mask, None)


class _SegAdapter(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, args):
        return self.net(args[0])


class SegmentationRecipe(_Recipe):
    """Stage 1: ``free_last_blocks`` >= 0 freezes all but the last ``free_last_blocks`` entries of ``net.encoder.features``
    (0 = the whole encoder) before the flat buffers are built; ``unfreeze()`` starts stage 2 (every parameter trainable,
    fresh momentum, the schedule restarted) -- the reference re-creates optimizer and scheduler between its stages."""

    def __init__(self, net, free_last_blocks=0, gamma=0, background_weights=1, words_weights=2, base_lr=1e-4, max_lr=4e-4,
                 step_size=2000, mode="triangular", momentum=0.9, weight_decay=1e-3, **trainer_kw):
        self.net = net
        self._cfg = dict(base_lr=base_lr, max_lr=max_lr, step_size=step_size, mode=mode)
        self._opt = dict(momentum=momentum, weight_decay=weight_decay, **trainer_kw)
        self.criterion = BinaryFocalLoss(gamma, background_weights, words_weights)
        self.stage = 1
        self._trainable = [p for p in net.parameters() if p.requires_grad]    # what stage 2 re-enables (never more than that)
        if free_last_blocks >= 0:
            if hasattr(net.encoder, "freeze_params"):
                net.encoder.freeze_params(free_last_blocks)
            elif free_last_blocks == 0:                      # Xception has no per-stage helper in the reference: all or nothing
                for p in net.encoder.parameters():
                    p.requires_grad_(False)
            else:
                raise ValueError("this encoder can only be frozen as a whole (free_last_blocks=0)")
        self._build()

    def _build(self):
        trainer = FlatSGDTrainer(_SegAdapter(self.net), lr=self._cfg["base_lr"], loss_fn=lambda out, target: self.criterion(out, target), **self._opt)
        _Recipe.__init__(self, trainer, **self._cfg)

    def unfreeze(self):
        for p in self._trainable:
            p.requires_grad_(True)
        self.stage = 2
        self.trainer.close()        # stage 1's gradient hooks and flat buffers go before stage 2's trainer is built
        self._build()
        return self

    def step(self, image, target):
        self.scheduler.batch_step()
        return self.trainer.step(image, None, target)

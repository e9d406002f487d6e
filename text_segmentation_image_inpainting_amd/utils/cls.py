"""Cyclical learning-rate schedule with the interface of the reference's ``models/utils/cls.py:10-157`` (row n3 of
SURVEY.md 8(f)): ``CyclicLR(optimizer, base_lr, max_lr, step_size, mode, gamma, scale_fn, scale_mode,
last_batch_iteration)``, ``batch_step()`` after every batch, ``get_lr()``.

Host-side only.  ``optimizer`` is anything with ``param_groups`` (a torch optimizer) or a
``train_step.FlatSGDTrainer`` (its single ``lr`` is driven).

    cycle = floor(1 + it / (2*step_size));  x = |it/step_size - 2*cycle + 1|
    lr    = base_lr + (max_lr - base_lr) * max(0, 1 - x) * scale(cycle or it)
    scale: triangular 1 | triangular2 1/2^(cycle-1) | exp_range gamma^it
"""
import math


class CyclicLR(object):
    def __init__(self, optimizer, base_lr=1e-3, max_lr=6e-3, step_size=2000, mode="triangular", gamma=1.0,
                 scale_fn=None, scale_mode="cycle", last_batch_iteration=-1):
        self.optimizer = optimizer
        self._groups = getattr(optimizer, "param_groups", None)
        n = len(self._groups) if self._groups is not None else 1
        self.base_lrs = self._per_group("base_lr", base_lr, n)
        self.max_lrs = self._per_group("max_lr", max_lr, n)
        self.step_size = step_size
        if mode not in ("triangular", "triangular2", "exp_range") and scale_fn is None:
            raise ValueError("mode is invalid and scale_fn is None")
        self.mode, self.gamma = mode, gamma
        if scale_fn is None:
            self.scale_fn, self.scale_mode = {
                "triangular": (lambda x: 1.0, "cycle"),
                "triangular2": (lambda x: 1.0 / (2.0 ** (x - 1)), "cycle"),
                "exp_range": (lambda x: self.gamma ** x, "iterations"),
            }[mode]
        else:
            self.scale_fn, self.scale_mode = scale_fn, scale_mode
        self.batch_step(last_batch_iteration + 1)
        self.last_batch_iteration = last_batch_iteration

    @staticmethod
    def _per_group(name, value, n):
        if isinstance(value, (list, tuple)):
            if len(value) != n:
                raise ValueError("Expected {} {}, got {}".format(n, name, len(value)))
            return list(value)
        return [value] * n

    def get_lr(self):
        it = self.last_batch_iteration
        cycle = math.floor(1 + it / (2 * self.step_size))
        x = abs(it / self.step_size - 2 * cycle + 1)
        scale = self.scale_fn(cycle if self.scale_mode == "cycle" else it)
        return [lo + (hi - lo) * max(0.0, 1.0 - x) * scale for lo, hi in zip(self.base_lrs, self.max_lrs)]

    def batch_step(self, batch_iteration=None):
        if batch_iteration is None:
            batch_iteration = self.last_batch_iteration + 1
        self.last_batch_iteration = batch_iteration
        lrs = self.get_lr()
        if self._groups is not None:
            for group, lr in zip(self._groups, lrs):
                group["lr"] = lr
        else:
            self.optimizer.lr = lrs[0]

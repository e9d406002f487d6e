"""Memory savers: activation recomputation for the encoders.

The reference shrinks training memory two ways: ``InPlaceABN`` (models/partial_convolution.py:12-17,187-191 -- not
vendored, so its fallback runs) and ``MobileNetV2.forward_checkpoint`` (models/MobileNetV2.py:109-111,
``torch.utils.checkpoint`` around the encoder).  Here:

* inside a ``Conv -> BatchNorm -> act -> Conv`` chain the normalised activation is never stored at all (K6b: the
  consumer re-applies (scale, shift, act) to the raw conv output while loading, forward AND backward), which is what
  InPlaceABN buys;
* ``checkpoint_sequential_stages`` recomputes whole stages in backward: only the tensors at the stage boundaries stay
  resident between forward and backward.  Unlike a plain ``torch.utils.checkpoint`` the recomputation pass does NOT
  touch the BatchNorm running statistics / batch counters a second time (``recomputing()`` tells the BatchNorm ops),
  so a checkpointed step leaves the module in exactly the state of a plain step.
"""
import contextlib

import torch

_RECOMPUTING = False


def recomputing() -> bool:
    """True while a checkpointed segment is re-run in backward: BatchNorm then uses batch statistics as in the
    first pass but leaves running_mean / running_var / num_batches_tracked alone."""
    return _RECOMPUTING


@contextlib.contextmanager
def _recompute_pass():
    global _RECOMPUTING
    prev, _RECOMPUTING = _RECOMPUTING, True
    try:
        yield
    finally:
        _RECOMPUTING = prev


class _Segment(torch.autograd.Function):
    """One recomputed segment.  forward runs without a tape and keeps only the input; backward re-runs the segment
    with a tape, back-propagates through it (parameter gradients accumulate into ``.grad`` like in any backward)
    and returns the input gradient."""

    @staticmethod
    def forward(ctx, run, x, dummy):
        ctx.run = run
        ctx.save_for_backward(x)
        with torch.no_grad():
            return run(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        if not ctx.needs_input_grad[1] and not _any_trainable(ctx.run):
            return None, None, None        # nothing behind this segment wants a gradient (checkpoint_segment avoids such nodes)
        # the recomputation sees the input exactly as the first pass did (a data tensor stays a data tensor: kernels
        # may pick a different -- equally valid, differently rounded -- form when an input gradient is wanted)
        xin = x.detach().requires_grad_(ctx.needs_input_grad[1])
        with torch.enable_grad(), _recompute_pass():
            y = ctx.run(xin)
        if y.requires_grad:
            torch.autograd.backward(y, gy)
        return None, xin.grad, None


def _any_trainable(run) -> bool:
    """Does the stage own a parameter that wants a gradient?  (Callables that are not modules count as trainable.)"""
    params = getattr(run, "parameters", None)
    if params is None:
        return True
    return any(p.requires_grad for p in params())


def checkpoint_segment(run, x):
    """``run(x)`` with its activations recomputed in backward.  ``run`` must be a pure function of ``x`` and of
    module parameters / buffers (deterministic kernels: the recomputation reproduces the first pass bit for bit)."""
    if not torch.is_grad_enabled():
        return run(x)
    if not x.requires_grad and not _any_trainable(run):
        # a frozen stage fed by data (two-stage recipes freeze the encoder's first stages): no node at all -- its output
        # must not require grad either, or the segments behind it would back-propagate through the frozen weights for nothing
        with torch.no_grad():
            return run(x)
    # the dummy input carries requires_grad so the node is kept even when x itself needs no gradient (first stage)
    dummy = torch.empty(0, device=x.device, requires_grad=True)
    return _Segment.apply(run, x, dummy)


def checkpoint_sequential_stages(stages, x):
    """Chain of modules, one recomputed segment each."""
    for stage in stages:
        x = checkpoint_segment(stage, x)
    return x

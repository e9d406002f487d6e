"""Host-side mirror of the reference's base wrappers (models/BaseModels.py).

Same public surface -- ``BaseModule`` (tolerant ``load_state_dict``, ``initialize_weights``,
``set_activation_inplace``, ``total_parameters``) -- so checkpoints and calling code written for
the reference keep working; the arithmetic itself lives in the HIP kernels (see ops.py).
"""
from contextlib import contextmanager

from typing import NamedTuple, Optional

import torch
from torch import nn

from . import ops


class BaseModule(nn.Module):
    """Base class of every mirrored network: the reference's public helpers (models/BaseModels.py:12-71) --
    ``initialize_weights``, the tolerant ``load_state_dict``, ``set_activation_inplace``, ``total_parameters`` --
    with the same names and call signatures."""

    def __init__(self):
        super().__init__()
        self.act_fn = None

    def initialize_weights(self):
        """Kaiming-normal (fan_out, leaky_relu gain) for trainable convolutions with zeroed biases; BatchNorm
        affine parameters to (1, 0).  Frozen tensors (the all-ones ``mask_conv`` weights) are left alone."""
        for mod in self.modules():
            weight = getattr(mod, "weight", None)
            if weight is None or not weight.requires_grad:
                continue
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(weight, mode="fan_out", nonlinearity="leaky_relu")
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.ones_(weight)
                nn.init.zeros_(mod.bias)

    def load_state_dict(self, state_dict, strict=True, self_state=False):
        """Copy-by-name loader that never raises (the reference's checkpoints are loaded into nets whose heads
        changed): entries with a matching name and shape are copied, everything else is reported and skipped.
        ``self_state`` may supply the destination mapping.  Returns (missing_in_model, failed) name lists."""
        target = self_state if self_state else self.state_dict()
        unknown, failed = [], []
        for key, value in state_dict.items():
            dst = target.get(key)
            if dst is None:
                unknown.append(key)
                continue
            try:
                dst.copy_(getattr(value, "data", value))
            except Exception as err:  # noqa: BLE001 - tolerance is the contract here
                failed.append(key)
                print(f"[load_state_dict] {key}: not loaded ({err})")
        for key in unknown:
            print(f"[load_state_dict] {key}: no entry of that name in this model")
        return unknown, failed

    @contextmanager
    def set_activation_inplace(self):
        """Switch the shared activation module to in-place for the duration of the block (memory saver of the
        reference's checkpointed forward; a no-op for the HIP kernels, which never keep the activation output)."""
        act = getattr(self, "act_fn", None)
        flip = act is not None and hasattr(act, "inplace")
        if flip:
            act.inplace = True
        try:
            yield
        finally:
            if flip:
                act.inplace = False

    def total_parameters(self):
        counts = [(p.numel(), p.requires_grad) for p in self.parameters()]
        total, trainable = sum(n for n, _ in counts), sum(n for n, t in counts if t)
        print(f"parameters: {total} total, {trainable} trainable")
        return total

    def forward(self, *x):
        raise NotImplementedError


# ---- layout / dispatch helpers shared by the mirrored modules -----------------------------
def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W]-shaped tensor (any memory format) -> NHWC-contiguous [N,H,W,C]; free for channels_last."""
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(y: torch.Tensor) -> torch.Tensor:
    """NHWC-contiguous -> [N,C,H,W]-shaped view (channels_last memory), as callers of the reference expect."""
    return y.permute(0, 3, 1, 2)


# num_batches_tracked bookkeeping: eager ``add_(1)`` per BatchNorm call (one tiny kernel each: 105 per ImageFill
# forward) unless a trainer batches them -- inside ``deferred_batch_counters()`` the increments are tallied on the host
# and applied with one multi-tensor add per distinct count when the block exits.
_DEFERRED_COUNTS = None


@contextmanager
def deferred_batch_counters():
    global _DEFERRED_COUNTS
    if _DEFERRED_COUNTS is not None:          # nested: the outer block flushes
        yield
        return
    _DEFERRED_COUNTS = {}
    try:
        yield
    finally:
        pending, _DEFERRED_COUNTS = _DEFERRED_COUNTS, None
        by_count = {}
        for tensor, n in pending.values():
            by_count.setdefault(n, []).append(tensor)
        for n, tensors in by_count.items():
            torch._foreach_add_(tensors, n)


def bn_state(bn):
    """(training, momentum, running_mean, running_var) for one BatchNorm2d call, and the batch-counter increment.
    While a checkpointed segment is recomputed (memory.recomputing()) the running statistics and the counter are
    left alone: batch statistics as in the first pass, no second update."""
    from .memory import recomputing
    training = bn.training or bn.running_mean is None
    if not training:
        return False, 0.0, bn.running_mean, bn.running_var
    momentum = bn_momentum(bn)
    if recomputing():
        return True, momentum, None, None
    counter = bn.num_batches_tracked
    if counter is not None:
        if _DEFERRED_COUNTS is None:
            counter.add_(1)
        else:
            rec = _DEFERRED_COUNTS.setdefault(id(counter), [counter, 0])
            rec[1] += 1
    return True, momentum, bn.running_mean, bn.running_var


def bn_momentum(bn) -> float:
    """BatchNorm2d.momentum; ``None`` (cumulative moving average) has no kernel -- fail loudly instead of
    silently substituting torch's default."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) has no HIP kernel")
    return float(bn.momentum)


def act_code(act):
    """nn activation module -> (kernel activation id, slope)."""
    if act is None or act is False:
        return ops.ACT_NONE, 0.0
    if isinstance(act, nn.LeakyReLU):
        return ops.ACT_LEAKY, float(act.negative_slope)
    if isinstance(act, nn.ReLU6):
        return ops.ACT_RELU6, 0.0
    if isinstance(act, nn.ReLU):
        return ops.ACT_RELU, 0.0
    raise NotImplementedError(f"activation {act!r} has no HIP kernel (ReLU / ReLU6 / LeakyReLU only)")


# +++++++++++++++++++++++++++++++++++++
#           Convolution Wrappers   (models/BaseModels.py:91-127)
# -------------------------------------
# The reference builds its nets from torch.nn building blocks inside nn.Sequential containers; the
# subclasses below keep the parameter names / container indices (so state_dict keys match) and route
# forward() to the HIP kernels.  Tensors between modules are NCHW-shaped with channels_last memory,
# so the NHWC views the kernels want are free.
class Conv2d(nn.Conv2d):
    bn_follows = False    # set by Conv_block: the conv also emits the BatchNorm statistics partials of its output (K6b)

    def forward(self, x):
        if self.padding_mode != "zeros":
            raise NotImplementedError("only zero padding has a HIP kernel")
        g = ops.make_geom(self.kernel_size, self.stride, self.padding, self.dilation)
        if self.bn_follows and self.training:
            y, part = ops.conv2d(to_nhwc(x), self.weight, self.bias, g, self.groups, want_stats=True)
            out = to_nchw(y)
            if part is not None:           # picked up by the BNAct that nn.Sequential calls next with this very object
                out._tsii_stat_part = (part, out.data_ptr(), out._version)
            return out
        return to_nchw(ops.conv2d(to_nhwc(x), self.weight, self.bias, g, self.groups))


class BNAct(nn.Sequential):
    """nn.Sequential(nn.BatchNorm2d(c)[, activation]) (models/BaseModels.py:95-99) as ONE fused kernel."""

    def forward(self, x, residual=None):
        bn = self[0]
        act, slope = act_code(self[1] if len(self) > 1 else None)
        training, momentum, rmean, rvar = bn_state(bn)
        res = None if residual is None else to_nhwc(residual)
        # statistics partials the producing Conv2d left on this very tensor object (K6b); used only while the tensor
        # still is that conv's untouched output (same storage, same version counter), otherwise the separate pass runs
        part = None
        tag = getattr(x, "_tsii_stat_part", None) if training else None
        if tag is not None and tag[1] == x.data_ptr() and tag[2] == x._version and tag[0].shape[-1] == x.shape[1]:
            part = tag[0]
        y = ops.bn_act(to_nhwc(x), bn.weight, bn.bias, rmean, rvar, training, momentum, bn.eps, act, slope, res, part)
        return to_nchw(y)


class Activation(nn.Module):
    """A bare activation placed in a Sequential (models/BaseModels.py:100-101)."""

    def __init__(self, act):
        super().__init__()
        self.act = act

    def forward(self, x):
        code, slope = act_code(self.act)
        return to_nchw(ops.activation(to_nhwc(x), code, slope))


class AvgPool2d(nn.AvgPool2d):
    def forward(self, x):
        k, s, p = self.kernel_size, self.stride, self.padding
        if not (isinstance(k, int) and isinstance(s, int) and isinstance(p, int)) or self.ceil_mode or not self.count_include_pad:
            raise NotImplementedError("only square AvgPool2d with count_include_pad=True has a HIP kernel")
        return to_nchw(ops.avg_pool(to_nhwc(x), k, s, p))


class Upsample(nn.Upsample):
    def forward(self, x):
        if self.mode != "bilinear" or self.align_corners or int(self.scale_factor) != self.scale_factor:
            raise NotImplementedError("only integer-factor bilinear (align_corners=False) up-sampling has a HIP kernel")
        return to_nchw(ops.bilinear_up(to_nhwc(x), int(self.scale_factor)))


class PixelShuffle(nn.PixelShuffle):
    """nn.PixelShuffle on the HIP path (no counterpart in the reference's code -- only its README)."""

    def forward(self, x):
        return to_nchw(ops.pixel_shuffle(to_nhwc(x), self.upscale_factor))


def interpolate_bilinear(x, scale_factor):
    """F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False)."""
    return to_nchw(ops.bilinear_up(to_nhwc(x), int(scale_factor)))


def cat_channels(xs):
    """torch.cat(xs, dim=1) on NCHW-shaped tensors."""
    return to_nchw(ops.concat([to_nhwc(x) for x in xs]))


class ConvSpec(NamedTuple):
    """One ``conv [-> BatchNorm [-> activation]]`` piece of a network table.  ``groups="dw"`` = depth-wise
    (groups = input channels); ``out=None`` keeps the channel count."""
    out: Optional[int]
    kernel: object = 1
    stride: int = 1
    padding: object = 0
    dilation: int = 1
    groups: object = 1
    bias: bool = False
    bn: bool = True
    act: bool = True


def conv_pieces(cin, spec: ConvSpec, activation):
    """Modules of one ConvSpec in the reference's container layout (models/BaseModels.py:91-102): the conv, then
    ``BNAct(BatchNorm2d[, act])`` if normalised, else the bare activation -- this index layout IS the state_dict key
    layout.  Returns (modules, output channels)."""
    cout = cin if spec.out is None else spec.out
    groups = cin if spec.groups == "dw" else spec.groups
    conv = Conv2d(cin, cout, spec.kernel, spec.stride, spec.padding, spec.dilation, groups, spec.bias)
    conv.bn_follows = bool(spec.bn)
    act = activation if spec.act else None
    mods = [conv]
    if spec.bn:
        mods.append(BNAct(nn.BatchNorm2d(cout), act) if act else BNAct(nn.BatchNorm2d(cout)))
    elif act is not None:
        mods.append(Activation(act))
    return mods, cout


def build_chain(cin, specs, activation):
    """Flat module list of a table of ConvSpecs (for ``nn.Sequential(*...)``) and its output channel count."""
    mods = []
    for spec in specs:
        piece, cin = conv_pieces(cin, spec, activation)
        mods += piece
    return mods, cin


def Conv_block(in_channels, out_channels, kernel_size, stride=1, padding=0,
               dilation=1, groups=1, bias=True, BN=False, activation=None):
    """The reference's list-returning factory (models/BaseModels.py:91-102), kept for code written against it."""
    spec = ConvSpec(out_channels, kernel_size, stride, padding, dilation, groups, bias, bool(BN), bool(activation))
    return conv_pieces(in_channels, spec, activation if activation else None)[0]


def run_chain(mods, x, final_residual=None):
    """Run a list of modules built from Conv_block pieces on an NCHW-shaped tensor with every ``Conv2d -> BNAct`` pair
    folded (K6b): the conv emits the BatchNorm statistics partials and, when another Conv2d follows, the normalised
    activation stays virtual (ops.LazyBN) and is applied by that conv while loading (models/MobileNetV2.py:127-140,
    models/BaseModels.py:105-127 chains).  Anything else in the list runs through its own forward().
    ``final_residual`` (NCHW-shaped; the list must end in a ``Conv2d -> BNAct`` pair): added by the pass that writes the last
    BatchNorm out, i.e. ``chain(x) + final_residual`` without a separate add pass (models/Xception.py:44)."""
    h, lazy, i, n = to_nhwc(x), None, 0, len(mods)
    if final_residual is not None:
        assert n >= 2 and isinstance(mods[-2], Conv2d) and isinstance(mods[-1], BNAct), "final_residual needs a closing conv + BatchNorm pair"
    while i < n:
        m = mods[i]
        if isinstance(m, Conv2d) and i + 1 < n and isinstance(mods[i + 1], BNAct) and m.padding_mode == "zeros":
            bn = mods[i + 1][0]
            act, slope = act_code(mods[i + 1][1] if len(mods[i + 1]) > 1 else None)
            training, momentum, rmean, rvar = bn_state(bn)
            g = ops.make_geom(m.kernel_size, m.stride, m.padding, m.dilation)
            src = lazy if lazy is not None else h
            if training:
                y, part = ops.conv2d(src, m.weight, m.bias, g, m.groups, want_stats=True)
            else:
                y, part = ops.conv2d(src, m.weight, m.bias, g, m.groups), None
            lz = ops.bn_lazy(y, bn.weight, bn.bias, rmean, rvar, training, momentum, bn.eps, act, slope, part)
            if i + 2 < n and isinstance(mods[i + 2], Conv2d) and ops.load_time_act(act, slope):
                h, lazy = None, lz
            else:
                res = to_nhwc(final_residual) if (final_residual is not None and i + 2 == n) else None
                h, lazy = lz.materialize(res), None
            i += 2
            continue
        if lazy is not None:
            h, lazy = lazy.materialize(), None
        h = to_nhwc(m(to_nchw(h)))
        i += 1
    if lazy is not None:
        h = lazy.materialize()
    return to_nchw(h)


class DSConvBlock(BaseModule):
    """depth-wise separable convolution (models/BaseModels.py:105-127)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, bias=True, BN=False, activation_dep=None, activation_point=None):
        super().__init__()
        self.depth_wise_conv = nn.Sequential(
            *Conv_block(in_channels, in_channels, kernel_size, stride, padding,
                        dilation, in_channels, bias, BN=BN, activation=activation_dep))
        self.point_wise_conv = nn.Sequential(
            *Conv_block(in_channels, out_channels, kernel_size=1, stride=1, padding=0,
                        dilation=1, bias=bias, BN=BN, activation=activation_point))

    def forward(self, x):
        return run_chain(list(self.depth_wise_conv) + list(self.point_wise_conv), x)


def run_nhwc(layer, x, mp):
    """Run a mirrored module / nn.Sequential of them on (NHWC tensor, MaskParts)."""
    if hasattr(layer, "forward_nhwc"):
        return layer.forward_nhwc(x, mp)
    if isinstance(layer, nn.Sequential):
        from .partial_convolution import PartialActivatedBN, PartialConv, PartialConv1x1, run_block
        if len(layer) == 2 and isinstance(layer[0], (PartialConv, PartialConv1x1)) and isinstance(layer[1], PartialActivatedBN):
            return run_block(layer, x, mp)      # conv + BatchNorm: statistics fused into the conv (K6b)
        for m in layer:
            x, mp = run_nhwc(m, x, mp)
        return x, mp
    raise NotImplementedError(f"{type(layer).__name__} is not part of the partial-convolution path")

"""Host-side mirror of the reference's base wrappers (models/BaseModels.py).

Same public surface -- ``BaseModule`` (tolerant ``load_state_dict``, ``initialize_weights``,
``set_activation_inplace``, ``total_parameters``) -- so checkpoints and calling code written for
the reference keep working; the arithmetic itself lives in the HIP kernels (see ops.py).
"""
import math
from contextlib import contextmanager

import torch
from torch import nn

from . import ops


class BaseModule(nn.Module):
    def __init__(self):
        self.act_fn = None
        super().__init__()

    def selu_init_params(self):  # models/BaseModels.py:17-29
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.weight.requires_grad:
                m.weight.data.normal_(0.0, 1.0 / math.sqrt(m.weight.numel()))
                if m.bias is not None:
                    m.bias.data.fill_(0)
            elif isinstance(m, nn.BatchNorm2d) and m.weight.requires_grad:
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear) and m.weight.requires_grad:
                m.weight.data.normal_(0, 1.0 / math.sqrt(m.weight.numel()))
                m.bias.data.zero_()

    def initialize_weights(self):  # models/BaseModels.py:31-39
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.weight.requires_grad:
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu")
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d) and m.weight.requires_grad:
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def load_state_dict(self, state_dict, strict=True, self_state=False):
        """Copy-by-name, print-and-continue loader (models/BaseModels.py:41-52): never raises."""
        own_state = self_state if self_state else self.state_dict()
        for name, param in state_dict.items():
            if name in own_state:
                try:
                    own_state[name].copy_(param.data)
                except Exception as e:  # noqa: BLE001 - mirrors the reference's tolerance
                    print("Parameter {} fails to load.".format(name))
                    print("-----------------------------------------")
                    print(e)
            else:
                print("Parameter {} is not in the model. ".format(name))

    @contextmanager
    def set_activation_inplace(self):  # models/BaseModels.py:54-62
        if hasattr(self, "act_fn") and hasattr(self.act_fn, "inplace"):
            self.act_fn.inplace = True
            yield
            self.act_fn.inplace = False
        else:
            yield

    def total_parameters(self):
        total = sum(i.numel() for i in self.parameters())
        trainable = sum(i.numel() for i in self.parameters() if i.requires_grad)
        print("Total parameters : {}. Trainable parameters : {}".format(total, trainable))
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


def run_nhwc(layer, x, mp):
    """Run a mirrored module / nn.Sequential of them on (NHWC tensor, MaskParts)."""
    if hasattr(layer, "forward_nhwc"):
        return layer.forward_nhwc(x, mp)
    if isinstance(layer, nn.Sequential):
        for m in layer:
            x, mp = run_nhwc(m, x, mp)
        return x, mp
    raise NotImplementedError(f"{type(layer).__name__} is not part of the partial-convolution path")

"""Mask bookkeeping for the partial-convolution path.

The reference carries masks as full ``[N, C, H, W]`` fp32 tensors and pushes them through
all-ones convolutions (models/partial_convolution.py:41-47,57-64).  After the first layer every
mask in the reference networks is constant across channels (or a concatenation of two such
masks in the decoder, models/image_inpainting.py:83-84), so here a mask is a short list of
*parts*: a channel-constant part is one ``[N, H, W]`` plane plus a channel count; a general
per-channel part (only ever the user's 3-channel input mask) keeps its NHWC tensor.
``as_tensor()`` reproduces the tensor the reference would have returned (value-equal; a
stride-0 expanded view where the reference itself returns one, :76-77,104).
"""
from typing import List, Optional

import torch

from . import ops


class Part:
    __slots__ = ("plane", "full", "channels", "premultiplied", "_sum")

    def __init__(self, channels: int, plane: Optional[torch.Tensor] = None, full: Optional[torch.Tensor] = None,
                 premultiplied: bool = False):
        assert (plane is None) != (full is None)
        self.plane, self.full, self.channels, self._sum = plane, full, int(channels), None
        # a general (per-channel) part whose feature channels were already multiplied by it
        # (ops.mul_mask) when the concatenated tensor was built: x*mask needs no further scaling
        self.premultiplied = bool(premultiplied)

    @property
    def planar(self):
        return self.plane is not None

    def count_plane(self):
        """(plane, multiplicity) whose product is this part's contribution to sum_c mask."""
        if self.planar:
            return self.plane, float(self.channels)
        if self._sum is None:
            self._sum = ops.mask_channel_sum(self.full.permute(0, 3, 1, 2))
        return self._sum, 1.0

    def first_channel_plane(self):
        if self.planar:
            return self.plane
        return ops.mask_channel_sum(self.full.permute(0, 3, 1, 2), channels=1)


class MaskParts:
    """A mask over C channels made of 1..2 parts (see module docstring)."""

    def __init__(self, parts: List[Part]):
        assert 1 <= len(parts) <= 2, "masks are one part, or two after a decoder concat"
        self.parts = parts

    @property
    def channels(self):
        return sum(p.channels for p in self.parts)

    @property
    def planar(self):
        return all(p.planar for p in self.parts)

    @property
    def fusable(self):
        """x*mask can ride in the conv kernels' row scale (planes; a trailing premultiplied part)."""
        ps = self.parts
        if len(ps) == 1:
            return ps[0].planar or ps[0].premultiplied
        return ps[0].planar and (ps[1].planar or ps[1].premultiplied)

    @staticmethod
    def from_plane(plane: torch.Tensor, channels: int) -> "MaskParts":
        return MaskParts([Part(channels, plane=plane)])

    @staticmethod
    def from_tensor(mask: torch.Tensor) -> "MaskParts":
        """[N,C,H,W] tensor -> parts.  A stride-0 channel dim (expanded plane) or C == 1 is planar."""
        assert mask.dim() == 4
        n, c, h, w = mask.shape
        if c == 1 or mask.stride(1) == 0:
            plane = mask[:, 0].contiguous()
            return MaskParts([Part(c, plane=plane)])
        return MaskParts([Part(c, full=mask.permute(0, 2, 3, 1).contiguous())])

    def cat(self, other: "MaskParts") -> "MaskParts":
        assert len(self.parts) == 1 and len(other.parts) == 1
        return MaskParts([self.parts[0], other.parts[0]])

    def upsample2x(self) -> "MaskParts":
        assert self.planar and len(self.parts) == 1
        p = self.parts[0]
        return MaskParts([Part(p.channels, plane=ops.plane_upsample2x(p.plane))])

    def row_scale(self):
        """(r0, split, r1) for the fused x*mask of planar masks."""
        assert self.fusable
        p0 = self.parts[0]
        if len(self.parts) == 1:
            return (p0.plane, p0.channels, None) if p0.planar else (None, 0, None)
        return p0.plane, p0.channels, self.parts[1].plane  # plane None (premultiplied) -> scale 1

    def count_operands(self):
        """(p0, a0, p1, a1): sum_c mask = a0*p0 + a1*p1 (exact small integers)."""
        p0, a0 = self.parts[0].count_plane()
        if len(self.parts) == 1:
            return p0, a0, None, 0.0
        p1, a1 = self.parts[1].count_plane()
        return p0, a0, p1, a1

    def first_channel_plane(self):
        return self.parts[0].first_channel_plane()

    def full_nhwc(self) -> torch.Tensor:
        """Materialised [N,H,W,C] mask (only for general masks on the slow, general path)."""
        outs = []
        for p in self.parts:
            outs.append(p.full if not p.planar else p.plane.unsqueeze(-1).expand(-1, -1, -1, p.channels))
        return outs[0].contiguous() if len(outs) == 1 else torch.cat(outs, dim=-1)

    def as_tensor(self) -> torch.Tensor:
        """The [N,C,H,W] tensor the reference would hand back."""
        outs = []
        for p in self.parts:
            if p.planar:
                outs.append(p.plane.unsqueeze(1).expand(-1, p.channels, -1, -1))
            else:
                outs.append(p.full.permute(0, 3, 1, 2))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)


def as_parts(mask) -> MaskParts:
    return mask if isinstance(mask, MaskParts) else MaskParts.from_tensor(mask)

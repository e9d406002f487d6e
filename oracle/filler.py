"""Deterministic, name-keyed weight filler (test infrastructure).

Both the golden generator (which fills the *reference* modules) and the tests
(which fill the product modules and feed the oracle) use this, so the three
see identical parameters without shipping any weights in the fixtures.

Scaling rationale: with the reference's default init the conv
term of a partial convolution is O(1e-3) and the output is ~bias, which hides
errors.  Here feature-conv weights are drawn so the conv term is O(1) and
BatchNorm affine/running statistics are non-trivial.
"""
import zlib

import numpy as np
import torch


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.default_rng((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)


def fill_tensor(key: str, shape, seed: int = 0, gain: float = 2.0) -> np.ndarray:
    """Value for state_dict entry ``key`` of ``shape`` (float32 / int64)."""
    r = _rng(key, seed)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if key.endswith("mask_conv.weight"):
        return np.ones(shape, dtype=np.float32)  # frozen all-ones (partial_convolution.py:44)
    if key.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if key.endswith("running_var"):
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith(".bias") and len(shape) == 1:
        return (0.2 * r.standard_normal(shape)).astype(np.float32)
    if key.endswith(".weight") and len(shape) == 1:  # BN gamma
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith(".weight") and len(shape) >= 2:  # conv / linear
        fan_in = int(np.prod(shape[1:]))
        std = (gain / max(fan_in, 1)) ** 0.5   # gain 2 = He init; deep eval-mode nets use 1 to stay O(1)
        return (std * r.standard_normal(shape)).astype(np.float32)
    return (0.1 * r.standard_normal(shape)).astype(np.float32)


def fill_state_dict_(state_dict, seed: int = 0, gain: float = 2.0):
    """In-place fill of every tensor of a ``state_dict`` (reference or product)."""
    with torch.no_grad():
        for k, v in state_dict.items():
            val = torch.from_numpy(fill_tensor(k, v.shape, seed, gain))
            v.copy_(val.to(v.dtype).reshape(v.shape))
    return state_dict


def make_state_dict(key_shapes, seed: int = 0, dtype=torch.float32, gain: float = 2.0):
    """Build a fresh filled state_dict from ``[(key, shape), ...]``."""
    out = {}
    for k, shape in key_shapes:
        t = torch.from_numpy(fill_tensor(k, shape, seed, gain))
        out[k] = t if t.dtype == torch.int64 else t.to(dtype)
    return out


def seeded_input(n, c, h, w, seed=0, hole_frac=0.25, per_channel_mask=False, blocky=True):
    """Seeded (image, mask) pair.  mask: 1 = valid, 0 = hole (float32 0/1)."""
    r = np.random.default_rng(1000 + seed)
    x = r.standard_normal((n, c, h, w)).astype(np.float32)
    mc = c if per_channel_mask else 1
    if blocky:
        # rectangular holes so that some conv windows are entirely inside a hole
        m = np.ones((n, mc, h, w), dtype=np.float32)
        for i in range(n):
            for j in range(mc):
                for _ in range(max(1, int(hole_frac * 8))):
                    hh = int(r.integers(2, max(3, h // 3)))
                    ww = int(r.integers(2, max(3, w // 3)))
                    y0 = int(r.integers(0, h - hh + 1))
                    x0 = int(r.integers(0, w - ww + 1))
                    m[i, j, y0:y0 + hh, x0:x0 + ww] = 0.0
    else:
        m = (r.uniform(size=(n, mc, h, w)) > hole_frac).astype(np.float32)
    if not per_channel_mask:
        m = np.repeat(m, c, axis=1)
    return torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(m))

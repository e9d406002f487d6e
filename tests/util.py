"""Shared parity metric: max-normalised relative error."""
import numpy as np
import torch


def to_np(a):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().double().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_err(a, b, floor=1e-30):
    """max|a-b| / max(max|b|, floor) over finite entries of b; NaN pattern must agree exactly.
    ``floor`` keeps analytically-zero quantities (both sides rounding noise) from failing."""
    a, b = to_np(a), to_np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert np.array_equal(fa, fb), "finite/NaN pattern differs: %d vs %d non-finite" % ((~fa).sum(), (~fb).sum())
    if not fb.any():
        return 0.0
    scale = np.abs(b[fb]).max()
    return float(np.abs(a[fb] - b[fb]).max() / max(scale, floor))


def assert_close(a, b, tol=1e-3, what="", floor=1e-30):
    e = rel_err(a, b, floor)
    assert e <= tol, f"{what}: max-normalised rel err {e:.3e} > {tol:.1e}"
    return e

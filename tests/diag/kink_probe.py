#!/usr/bin/env python3
"""Diagnostic behind tests/util.py:assert_gradients_close: for the ImageFill 64^2 train step, list the gradient tensors
furthest from the oracle and test whether their error is (near) rank-1 = the contribution of single pixels whose activation
sits on the other side of a LeakyReLU kink, as opposed to the dense error pattern of an inaccurate kernel.
    python tests/diag/kink_probe.py            (GPU box; the oracle runs on the host cores)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from tests.test_parity_ops import _imagefill_case
    from tests.util import rel_err
    dev = torch.device("cuda:0")
    model, sd, y, yo, loss, lo = _imagefill_case(dev, 64, 2, 66, True)
    rows = sorted(((rel_err(p.grad, sd[k].grad, 1e-6), k) for k, p in model.named_parameters() if p.requires_grad), reverse=True)
    print("output rel err %.2e, loss %.7f vs %.7f" % (rel_err(y, yo), float(loss), float(lo)))
    for e, k in rows[:6]:
        g, r = dict(model.named_parameters())[k].grad.detach().cpu().double(), sd[k].grad.double()
        d = (g - r).reshape(g.shape[0], -1)
        line = "%.3e  %-48s" % (e, k)
        if d.shape[0] > 1 and d.shape[1] > 1:
            sv = torch.linalg.svdvals(d)
            line += "  error matrix %dx%d: top singular value carries %.1f %% of the Frobenius norm, top 3: %.1f %%" % (
                d.shape[0], d.shape[1], 100 * float(sv[0] ** 2 / (sv ** 2).sum()), 100 * float((sv[:3] ** 2).sum() / (sv ** 2).sum()))
        else:
            line += "  (vector) largest entry carries %.1f %% of the squared error" % (100 * float((d ** 2).max() / (d ** 2).sum()))
        print(line)
    print("median over %d tensors: %.2e" % (len(rows), float(np.median([e for e, _ in rows]))))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Diagnostic behind tests/test_training_recipe.py (segmentation recipe): TextSegament(width_mult=2) on the 64x64 fixture input,
ONE forward + backward, every decoder-side gradient against the fp64 oracle in each arithmetic mode (6 = split bf16, 0 = f32
MFMA): error relative to the tensor's largest entry AND relative to what decides an SGD update at lr 1e-4 (the recipe test's
yardstick), with the rank structure of the error for the worst tensors.
    python tests/diag/seg_grad_probe.py            (GPU box; the oracle runs on the host cores)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd import _lib
    from oracle import seg_oracle as S
    from oracle.filler import fill_state_dict_, make_state_dict
    from tests.util import low_rank_error
    G = np.load(os.path.join(ROOT, "tests", "golden", "textsegament_64.npz"))
    x, t = torch.from_numpy(G["x"]), torch.from_numpy(G["t"])
    probe = T.TextSegament(width_mult=2)
    keys = [(k, tuple(v.shape)) for k, v in probe.state_dict().items()]
    names = [k for k, _ in probe.named_parameters() if not k.startswith("encoder.")]

    def oracle(dtype):
        sd = make_state_dict(keys, seed=41, gain=1.0, dtype=dtype)
        for k in names:
            sd[k].requires_grad_(True)
        loss = S.binary_focal_loss(S.text_segament(sd, x.to(dtype), training=True, width_mult=2), t.to(dtype), 0.0, 1.0, 2.0)
        loss.backward()
        return {k: sd[k].grad.double() for k in names}, float(loss)
    g64, l64 = oracle(torch.float64)
    g32, l32 = oracle(torch.float32)
    dev = torch.device("cuda:0")
    for mode in (6, 0):
        _lib.set_gemm_products(mode)
        net = T.TextSegament(width_mult=2)
        fill_state_dict_(net.state_dict(), seed=41, gain=1.0)
        net = net.to(dev).train()
        for p in net.encoder.parameters():
            p.requires_grad_(False)
        loss = T.BinaryFocalLoss(0, 1, 2)(net(x.to(dev)), t.to(dev))
        loss.backward()
        params = dict(net.named_parameters())
        rows = []
        for k in names:
            g = params[k].grad.detach().cpu().double()
            scale = float(g64[k].abs().max())
            rows.append((float((g - g64[k]).abs().max()) / max(scale, 1e-30), k, float((g32[k] - g64[k]).abs().max()) / max(scale, 1e-30), scale))
        rows.sort(reverse=True)
        print(f"mode {mode}: loss {float(loss):.7f} (oracle fp64 {l64:.7f}, fp32 {l32:.7f}); worst of {len(rows)} decoder-side gradient tensors")
        for e, k, n, scale in rows[:8]:
            ok, f = low_rank_error(params[k].grad.detach().cpu(), g64[k])
            print(f"   {e:.2e}  (oracle fp32-vs-fp64 {n:.2e}; max|g| {scale:.2e}; {100 * f:.1f} % of the error in <= 3 singular values / entries)  {k} {tuple(g64[k].shape)}")
        print(f"   median {np.median([r[0] for r in rows]):.2e}")
    _lib.set_gemm_products(None)


if __name__ == "__main__":
    main()

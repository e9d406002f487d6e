#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box only ever
sees the committed ``*.npz`` / ``*.json`` outputs.  Nothing from the reference is
copied: the fixtures hold seeded inputs, the reference's outputs / gradients,
and state_dict key+shape lists.  Weights are NOT stored -- they are regenerated
by ``oracle.filler`` (name-keyed, deterministic).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True

from oracle.filler import fill_state_dict_, seeded_input  # noqa: E402

from models import partial_convolution as rpc  # noqa: E402  (reference)
from models import image_inpainting as rii  # noqa: E402  (reference)
from models.MobileNetV2 import PartialInvertedResidual  # noqa: E402  (reference)

torch.manual_seed(0)
torch.set_num_threads(8)

# ---------------------------------------------------------------------------
# 1. op-level cases for a1/a2/a3
# ---------------------------------------------------------------------------
# kind, cin, cout, k, s, p, d, groups, bias, same_holes, per_channel_mask, H
OP_CASES = [
    ("pconv", 3, 8, 7, 2, 3, 1, 1, True, False, True, 20),     # ImageFill stem flavour (per-channel mask)
    ("pconv", 3, 8, 7, 2, 3, 1, 1, True, True, False, 20),     # ImageFillOrigin stem flavour
    ("pconv", 8, 8, 3, 1, 1, 1, 8, False, True, False, 12),    # depth-wise same_holes (divide by cnt*Cin quirk)
    ("pconv", 8, 8, 3, 2, 1, 1, 8, False, True, False, 12),    # depth-wise stride 2
    ("pconv", 8, 8, 3, 1, 2, 2, 8, False, True, False, 12),    # depth-wise dilation 2
    ("pconv", 8, 8, 3, 1, 4, 4, 8, True, True, False, 12),     # depth-wise dilation 4 + bias
    ("pconv", 8, 8, 3, 1, 8, 8, 8, False, True, False, 16),    # depth-wise dilation 8
    ("pconv", 6, 4, 3, 1, 1, 1, 1, True, False, True, 10),     # dense non-same-holes, per-channel mask
    ("pconv", 6, 4, 5, 2, 2, 1, 1, False, True, False, 12),    # dense 5x5 s2 same_holes
    ("pconv", 7, 3, 3, 1, 1, 1, 1, True, False, False, 10),    # odd channels (35->3 flavour)
    ("pconv", 4, 6, 3, 1, 2, 2, 1, False, False, False, 10),   # dense dilated (V2 flavour)
    ("pconv", 4, 4, 1, 1, 0, 1, 1, True, False, True, 8),      # 1x1 through the general class
    ("pconv1x1", 8, 16, 1, 1, 0, 1, 1, False, False, False, 8),
    ("pconv1x1", 8, 4, 1, 1, 0, 1, 1, True, False, True, 8),
    ("noholes", 8, 16, 1, 1, 0, 1, 1, False, False, False, 8),  # two-plane masks -> see below
    ("noholes", 6, 4, 1, 1, 0, 1, 1, True, False, True, 8),     # per-channel mask, NaN where all-hole
    ("noholes", 4, 4, 3, 1, 1, 1, 1, False, False, False, 8),
]


def run_op_cases():
    out = {}
    meta = []
    for idx, (kind, cin, cout, k, s, p, d, g, bias, same, pcm, H) in enumerate(OP_CASES):
        if kind == "pconv":
            m = rpc.PartialConv(cin, cout, k, s, p, d, g, bias, same)
        elif kind == "pconv1x1":
            m = rpc.PartialConv1x1(cin, cout, k, s, p, d, g, bias)
        else:
            m = rpc.PartialConvNoHoles(cin, cout, k, s, p, d, g, bias)
        fill_state_dict_(m.state_dict(), seed=idx)
        x, mask = seeded_input(2, cin, H, H, seed=idx, hole_frac=0.3, per_channel_mask=pcm,
                               blocky=(kind != "noholes" or pcm))
        if kind == "noholes" and not pcm:
            # decoder flavour: first half of the channels carry an all-ones plane
            mask[:, : cin // 2] = 1.0
        if kind == "noholes" and pcm:
            mask[:, :, 2:4, 3:6] = 0.0  # an all-hole block: 0/0 -> NaN (quirk F6)
        x.requires_grad_(True)
        y, nm = m((x, mask))
        finite = torch.isfinite(y)
        gy = torch.from_numpy(np.random.default_rng(77 + idx).standard_normal(tuple(y.shape)).astype(np.float32))
        gy = torch.where(finite, gy, torch.zeros_like(gy))
        # NaN outputs (NoHoles all-hole windows) are excluded from the backward seed
        (torch.where(finite, y, torch.zeros_like(y)) * gy).sum().backward()
        pre = f"op{idx}."
        out[pre + "x"] = x.detach().numpy()
        out[pre + "mask"] = mask.numpy()
        out[pre + "y"] = y.detach().numpy()
        out[pre + "new_mask"] = np.ascontiguousarray(nm.detach().numpy())
        out[pre + "gy"] = gy.numpy()
        out[pre + "dx"] = x.grad.numpy()
        out[pre + "dw"] = m.feature_conv.weight.grad.numpy()
        if bias:
            out[pre + "db"] = m.feature_conv.bias.grad.numpy()
        meta.append(dict(idx=idx, kind=kind, cin=cin, cout=cout, k=k, s=s, p=p, d=d, groups=g, bias=bias,
                         same_holes=same, per_channel_mask=pcm, H=H,
                         keys=[[kk, list(v.shape)] for kk, v in m.state_dict().items()]))
    np.savez_compressed(os.path.join(HERE, "pconv_ops.npz"), **out)
    with open(os.path.join(HERE, "pconv_ops.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("pconv_ops:", len(meta), "cases")


# ---------------------------------------------------------------------------
# 2. PartialInvertedResidual blocks (a7) in train mode
# ---------------------------------------------------------------------------
# in_c, out_c, k, s, p, d, t, use_1, no_holes, same_holes, H, two_plane_mask
PIR_CASES = [
    (8, 16, 3, 2, 1, 1, 4, True, False, True, 12, False),    # encoder, stride 2
    (16, 16, 3, 1, 1, 1, 4, True, False, True, 12, False),   # encoder, residual
    (16, 16, 3, 1, 2, 2, 4, False, True, True, 12, False),   # dilated, NoHoles 1x1, residual
    (24, 8, 3, 1, 1, 1, 2, False, True, True, 12, True),     # decoder: concat of two planes
]


def run_pir_cases():
    out = {}
    meta = []
    for idx, (ic, oc, k, s, p, d, t, u1, nh, sh, H, two) in enumerate(PIR_CASES):
        act = torch.nn.LeakyReLU(0.3)
        m = PartialInvertedResidual(ic, oc, k, s, p, d, t, bias=False, BN=True, activation=act,
                                    use_1_conv=u1, no_holes_1_conv=nh, same_holes=sh)
        fill_state_dict_(m.state_dict(), seed=100 + idx)
        m.train()
        x, mask = seeded_input(2, ic, H, H, seed=100 + idx, hole_frac=0.15, blocky=True)
        if nh:
            # NoHoles layers only see masks whose channel-sum is > 0 in the reference nets
            if two:
                mask[:, : ic // 3] = 1.0
            else:
                mask[:] = 1.0
        x.requires_grad_(True)
        y, nm = m((x, mask))
        gy = torch.from_numpy(np.random.default_rng(177 + idx).standard_normal(tuple(y.shape)).astype(np.float32))
        (y * gy).sum().backward()
        pre = f"pir{idx}."
        out[pre + "x"] = x.detach().numpy()
        out[pre + "mask"] = mask.numpy()
        out[pre + "y"] = y.detach().numpy()
        out[pre + "new_mask"] = np.ascontiguousarray(nm.detach().numpy())
        out[pre + "gy"] = gy.numpy()
        out[pre + "dx"] = x.grad.numpy()
        sd = m.state_dict()
        for kk, v in m.named_parameters():
            if v.grad is not None:
                out[pre + "grad." + kk] = v.grad.numpy()
        for kk, v in sd.items():
            if "running_" in kk:
                out[pre + "buf." + kk] = v.numpy().copy()
        meta.append(dict(idx=idx, in_c=ic, out_c=oc, k=k, s=s, p=p, d=d, t=t, use_1_conv=u1,
                         no_holes_1_conv=nh, same_holes=sh, H=H, two_plane=two,
                         keys=[[kk, list(v.shape)] for kk, v in sd.items()]))
    np.savez_compressed(os.path.join(HERE, "pir_blocks.npz"), **out)
    with open(os.path.join(HERE, "pir_blocks.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("pir_blocks:", len(meta), "cases")


# ---------------------------------------------------------------------------
# 3. whole models
# ---------------------------------------------------------------------------
def grad_sample_keys(model):
    keys = [k for k, p in model.named_parameters() if p.requires_grad and p.numel() <= 40000]
    pick = keys[:5] + keys[len(keys) // 2: len(keys) // 2 + 5] + keys[-5:]
    return sorted(set(pick))


def run_models():
    keylists = {}
    for name in ("ImageFill", "ImageFillOrigin", "ImageFillOriginV2"):
        model = getattr(rii, name)()
        keylists[name] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        keylists[name + ".trainable"] = [k for k, p in model.named_parameters() if p.requires_grad]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keylists, f)
    print("state_dict_keys:", {k: len(v) for k, v in keylists.items()})

    # ImageFill 64x64 bs 2: train fwd+bwd (L1-mean loss) and eval fwd, fp32 and fp64
    out = {}
    model = rii.ImageFill()
    fill_state_dict_(model.state_dict(), seed=7)
    x, mask = seeded_input(2, 3, 64, 64, seed=7, hole_frac=0.12, per_channel_mask=True)
    clean = torch.from_numpy(np.random.default_rng(9).standard_normal((2, 3, 64, 64)).astype(np.float32))
    out["x"], out["mask"], out["clean"] = x.numpy(), mask.numpy(), clean.numpy()
    model.eval()
    with torch.no_grad():
        out["y_eval"] = model((x, mask)).numpy()
    model.train()
    y = model((x, mask))
    loss = (y - clean).abs().mean()
    loss.backward()
    out["y_train"] = y.detach().numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    params = dict(model.named_parameters())
    for k in grad_sample_keys(model):
        out["grad." + k] = params[k].grad.numpy()
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            if k.startswith("encoder.1.0.") or k.startswith("decoder.2.0."):
                out["buf." + k] = v.numpy().copy()
    # fp64 run of the same thing: the noise-floor reference
    model64 = rii.ImageFill().double()
    fill_state_dict_(model64.state_dict(), seed=7)
    model64.train()
    y64 = model64((x.double(), mask.double()))
    out["y_train_f64"] = y64.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "imagefill_64.npz"), **out)
    print("imagefill_64: loss", loss.item(), "max|y|", float(y.abs().max()),
          "fp32-vs-fp64", float((y.double() - y64).abs().max()))

    # ImageFillOrigin / V2 at 256x256 bs 1 (smallest size they accept), eval + train forward
    for name in ("ImageFillOrigin", "ImageFillOriginV2"):
        model = getattr(rii, name)()
        fill_state_dict_(model.state_dict(), seed=11)
        x, mask = seeded_input(1, 3, 256, 256, seed=11, hole_frac=0.08, per_channel_mask=False)
        o = {"x": x.numpy().astype(np.float16), "mask": np.packbits(mask[:, :1].numpy().astype(np.uint8))}
        x = torch.from_numpy(o["x"].astype(np.float32))  # inputs are stored as fp16 -> use the rounded values
        model.eval()
        with torch.no_grad():
            o["y_eval"] = model((x, mask)).numpy()
        # train-mode BN needs >1 value per channel at the 1x1 bottleneck -> batch 2 (image + its flip)
        model.train()
        x2 = torch.cat([x, x.flip(3)], 0)
        m2 = torch.cat([mask, mask.flip(3)], 0)
        with torch.no_grad():
            o["y_train_b2"] = model((x2, m2)).numpy()
        np.savez_compressed(os.path.join(HERE, name.lower() + "_256.npz"), **o)
        print(name, "max|y_eval|", float(np.abs(o["y_eval"]).max()), "max|y_train|", float(np.abs(o["y_train_b2"]).max()))


if __name__ == "__main__":
    run_op_cases()
    run_pir_cases()
    run_models()

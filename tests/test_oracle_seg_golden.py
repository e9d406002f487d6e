"""Pins the segmentation oracle (oracle/seg_oracle.py) against fixtures produced by the imported
reference (tests/golden/make_golden_seg.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import seg_oracle as S
from oracle.filler import make_state_dict
from oracle.pconv_oracle import leaky
from tests.util import assert_close

ACT = leaky(0.3)


def run_block(c, sd, x):
    kw, kind = c["kw"], c["kind"]
    if kind == "ir":
        return S.inverted_residual(sd, "", x, kw["in_channel"], kw["out_channel"], kw["stride"], kw["expand_ratio"],
                                   kw["dilation"], ACT, kw["add_sece"], True)
    if kind == "scse":
        return S.scse(sd, "", x, ACT)
    if kind == "rfb":
        return S.rfb(sd, "", x, kw["out_channel"], ACT, True)
    if kind == "asp":
        return S.asp(sd, "", x, ACT, tuple(kw["asp_rate"]), True)
    return S.residual_block(sd, "", x, kw["in_channels"], kw["out_channels"], kw["stride"], kw["padding"],
                            kw["dilation"], ACT, True)


def test_seg_blocks(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "seg_blocks.json")))
    G = np.load(os.path.join(golden_dir, "seg_blocks.npz"))
    for c in meta:
        i = c["idx"]
        pre = f"blk{i}."
        sd = make_state_dict([(k, s) for k, s in c["keys"]], seed=300 + i)
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        x = torch.from_numpy(G[pre + "x"]).requires_grad_(True)
        y = run_block(c, sd, x)
        assert_close(y, G[pre + "y"], 1e-5, f"blk{i} {c['kind']} y")
        (y * torch.from_numpy(G[pre + "gy"])).sum().backward()
        assert_close(x.grad, G[pre + "dx"], 1e-5, f"blk{i} dx")
        # analytically-zero gradients (a BN bias in front of another train-mode BN) are rounding noise:
        # floor the scale at 1e-4 of the block's largest gradient
        gmax = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith(pre + "grad."))
        for k in G.files:
            if k.startswith(pre + "grad."):
                assert_close(sd[k[len(pre) + 5:]].grad, G[k], 3e-4, k, floor=1e-4 * gmax)
            if k.startswith(pre + "buf."):
                assert_close(sd[k[len(pre) + 4:]], G[k], 1e-6, k)


def test_focal_loss(golden_dir):
    G = np.load(os.path.join(golden_dir, "focal_loss.npz"))
    t = torch.from_numpy(G["t"])
    for j in range(4):
        g, bw, ww = G[f"c{j}.cfg"]
        x = torch.from_numpy(G["x"]).requires_grad_(True)
        l = S.binary_focal_loss(x, t, g, bw, ww)
        assert abs(l.item() - float(G[f"c{j}.loss"])) < 1e-6
        l.backward()
        assert_close(x.grad, G[f"c{j}.dx"], 1e-5, f"focal c{j} dx")


@pytest.mark.parametrize("name", ["TextSegament", "XceptionTextSegment"])
def test_seg_nets_64(golden_dir, name):
    keys = json.load(open(os.path.join(golden_dir, "seg_state_dict_keys.json")))[name]
    G = np.load(os.path.join(golden_dir, name.lower() + "_64.npz"))
    x, t = torch.from_numpy(G["x"]), torch.from_numpy(G["t"])
    fn = S.SEG_MODELS[name]
    sd = make_state_dict([(k, s) for k, s in keys], seed=41, gain=1.0)
    with torch.no_grad():
        ye = fn(sd, x, training=False)
    assert_close(ye, G["y_eval"], 1e-5, name + " eval")
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    y = fn(sd, x, training=True)
    assert_close(y, G["y_train"], 1e-4, name + " train")
    l = S.binary_focal_loss(y, t, 0.0, 1.0, 2.0)
    assert abs(l.item() - float(G["loss"])) < 1e-5
    l.backward()
    n = 0
    gmax = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith("grad."))
    for k in G.files:
        if k.startswith("grad."):
            assert_close(sd[k[5:]].grad, G[k], 1e-3, k, floor=1e-4 * gmax)
            n += 1
    assert n >= 12


def test_inpainting_loss(golden_dir):
    G = np.load(os.path.join(golden_dir, "inpainting_loss.npz"))
    keys = json.loads(str(G["keys"]))
    sd = make_state_dict([(k, s) for k, s in keys], seed=77)
    gt, mask = torch.from_numpy(G["gt"]), torch.from_numpy(G["mask"])
    out = torch.from_numpy(G["out"]).requires_grad_(True)
    l = S.inpainting_loss(sd, gt * mask, mask, out, gt, width_mult=1, feature_range=3, training=True)
    assert abs(l.item() - float(G["loss"])) < 1e-5 * abs(float(G["loss"]))
    l.backward()
    assert_close(out.grad, G["dout"], 1e-4, "InpaintingLoss d/d(output)")


@pytest.mark.parametrize("name", ["TextSegament", "XceptionTextSegment"])
def test_seg_oracle_256_vs_reference_fixture(name, golden_dir):
    """cfg 1 size: the oracle's eval / train outputs and focal loss against the reference's own 256 x 256 run
    (tests/golden/make_golden_misc.py).  Forward only here (the backward at this size is the GPU suite's reference run)."""
    keys = json.load(open(os.path.join(golden_dir, "seg_state_dict_keys.json")))[name]
    G = np.load(os.path.join(golden_dir, name.lower() + "_256.npz"))
    x = torch.from_numpy(np.random.default_rng(int(G["seed_x"])).standard_normal((2, 3, 256, 256)).astype(np.float32))
    t = (torch.from_numpy(np.random.default_rng(int(G["seed_t"])).uniform(size=(2, 1, 256, 256))) > 0.8).float()
    sd = make_state_dict([(k, s) for k, s in keys], seed=43, gain=1.0)
    with torch.no_grad():
        ye = S.SEG_MODELS[name](sd, x, training=False)
        y = S.SEG_MODELS[name](sd, x, training=True)
        loss = S.binary_focal_loss(y, t, 0.0, 1.0, 2.0)
    assert_close(ye, G["y_eval"], 1e-4, name + " 256 eval")
    assert_close(y, G["y_train"], 1e-4, name + " 256 train")
    assert abs(float(loss) - float(G["loss"])) < 1e-5

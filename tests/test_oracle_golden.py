"""Pins the CPU oracle against fixtures produced by the imported reference
(tests/golden/make_golden.py).  Runs on CPU, no GPU, no /root/reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pconv_oracle as O
from oracle.filler import make_state_dict
from tests.util import assert_close, rel_err

TOL = 2e-6  # same math, same backend


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a, grad=False):
    t = torch.from_numpy(np.array(a))
    return t.requires_grad_(True) if grad else t


def test_op_cases(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "pconv_ops.json")))
    G = _load(golden_dir, "pconv_ops.npz")
    for c in meta:
        i = c["idx"]
        pre = f"op{i}."
        sd = make_state_dict([(k, s) for k, s in c["keys"]], seed=i)
        w = sd["feature_conv.weight"].requires_grad_(True)
        b = sd.get("feature_conv.bias")
        if b is not None:
            b.requires_grad_(True)
        x, mask = _t(G[pre + "x"], True), _t(G[pre + "mask"])
        if c["kind"] == "pconv":
            y, nm = O.partial_conv(x, mask, w, b, c["s"], c["p"], c["d"], c["groups"], c["same_holes"])
        elif c["kind"] == "pconv1x1":
            y, nm = O.partial_conv1x1(x, mask, w, b)
        else:
            y, nm = O.partial_conv_noholes(x, mask, w, b, c["s"], c["p"], c["d"])
        assert_close(y, G[pre + "y"], TOL, f"op{i} y")
        assert np.array_equal(nm.detach().numpy(), G[pre + "new_mask"]), f"op{i} new_mask not bit-exact"
        gy = _t(G[pre + "gy"])
        fin = torch.isfinite(y)
        (torch.where(fin, y, torch.zeros_like(y)) * gy).sum().backward()
        assert_close(x.grad, G[pre + "dx"], TOL, f"op{i} dx")
        assert_close(w.grad, G[pre + "dw"], TOL, f"op{i} dw")
        if b is not None:
            assert_close(b.grad, G[pre + "db"], TOL, f"op{i} db")


def test_noholes_nan_case_present(golden_dir):
    """Quirk F6: all-hole 1x1 windows give NaN in PartialConvNoHoles -- the fixture holds such a case."""
    G = _load(golden_dir, "pconv_ops.npz")
    assert np.isnan(G["op15.y"]).any()


def test_pir_blocks(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "pir_blocks.json")))
    G = _load(golden_dir, "pir_blocks.npz")
    act = O.leaky(0.3)
    for c in meta:
        i = c["idx"]
        pre = f"pir{i}."
        sd = make_state_dict([(k, s) for k, s in c["keys"]], seed=100 + i)
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running" not in k and "mask_conv" not in k:
                v.requires_grad_(True)
        x, mask = _t(G[pre + "x"], True), _t(G[pre + "mask"])
        y, nm = O.partial_inverted_residual(sd, "", x, mask, c["in_c"], c["out_c"], c["k"], c["s"], c["p"],
                                            c["d"], c["t"], act, c["use_1_conv"], c["no_holes_1_conv"],
                                            c["same_holes"], True)
        assert_close(y, G[pre + "y"], 1e-5, f"pir{i} y")
        assert np.array_equal(nm.detach().numpy(), G[pre + "new_mask"])
        (y * _t(G[pre + "gy"])).sum().backward()
        assert_close(x.grad, G[pre + "dx"], 1e-5, f"pir{i} dx")
        for k in G.files:
            if k.startswith(pre + "grad."):
                assert_close(sd[k[len(pre) + 5:]].grad, G[k], 1e-5, k)
            if k.startswith(pre + "buf."):
                assert_close(sd[k[len(pre) + 4:]], G[k], 1e-6, k)


def test_imagefill_64(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    G = _load(golden_dir, "imagefill_64.npz")
    x, mask, clean = _t(G["x"]), _t(G["mask"]), _t(G["clean"])
    sd = make_state_dict([(k, s) for k, s in keys["ImageFill"]], seed=7)
    with torch.no_grad():
        y_eval = O.image_fill(sd, x, mask, training=False)
    assert_close(y_eval, G["y_eval"], 1e-5, "ImageFill eval")
    for k in keys["ImageFill.trainable"]:
        sd[k].requires_grad_(True)
    y = O.image_fill(sd, x, mask, training=True)
    # conv-term parity: remove the output bias, which otherwise dominates
    b = sd["decoder.3.0.feature_conv.bias"].detach().view(1, 3, 1, 1)
    assert_close(y.detach() - b, torch.from_numpy(G["y_train"]) - b, 1e-4, "ImageFill train (bias removed)")
    loss = O.l1_mean(y, clean)
    assert abs(loss.item() - float(G["loss"])) < 1e-6
    loss.backward()
    n = 0
    for k in G.files:
        if k.startswith("grad."):
            assert_close(sd[k[5:]].grad, G[k], 2e-4, k)
            n += 1
        if k.startswith("buf."):
            assert_close(sd[k[4:]], G[k], 1e-5, k)
    assert n >= 10
    # fp32 vs the reference's fp64 run: the noise floor is far below the 1e-3 bar
    assert rel_err(y.detach() - b, torch.from_numpy(G["y_train_f64"]).float() - b) < 1e-4


@pytest.mark.parametrize("name,fn", [("ImageFillOrigin", O.image_fill_origin),
                                     ("ImageFillOriginV2", O.image_fill_origin_v2)])
def test_origin_models_256(golden_dir, name, fn):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    G = _load(golden_dir, name.lower() + "_256.npz")
    x = torch.from_numpy(G["x"].astype(np.float32))
    plane = np.unpackbits(G["mask"])[: 256 * 256].reshape(1, 1, 256, 256).astype(np.float32)
    mask = torch.from_numpy(np.repeat(plane, 3, axis=1))
    sd = make_state_dict([(k, s) for k, s in keys[name]], seed=11)
    with torch.no_grad():
        y = fn(sd, x, mask, training=False)
    b = sd["decoder.7.0.feature_conv.bias"].view(1, 3, 1, 1)
    assert_close(y - b, torch.from_numpy(G["y_eval"]) - b, 1e-4, name + " eval (bias removed)")
    with torch.no_grad():
        y2 = fn(sd, torch.cat([x, x.flip(3)]), torch.cat([mask, mask.flip(3)]), training=True)
    assert_close(y2 - b, torch.from_numpy(G["y_train_b2"]) - b, 1e-3, name + " train b2 (bias removed)")

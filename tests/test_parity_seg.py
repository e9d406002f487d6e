"""Parity of the HIP segmentation path with reference-generated fixtures (tests/golden/make_golden_seg.py):
building blocks (InvertedResidual+scSE, scSE, RFB, ASP, Xception ResidualBlock), BinaryFocalLoss, and the two
whole nets.  1e-3 max-normalised tolerance."""
import json
import os

import numpy as np
import pytest
import torch

import text_segmentation_image_inpainting_amd as T
from oracle.filler import fill_state_dict_
from tests.backends import BACKENDS, both_backends
from tests.util import assert_close

TOL = 1e-3
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def build_block(c):
    act = torch.nn.LeakyReLU(0.3)
    kw, kind = dict(c["kw"]), c["kind"]
    if kind == "ir":
        return T.InvertedResidual(activation=act, bias=False, **kw)
    if kind == "scse":
        return T.SpatialChannelSqueezeExcitation(kw["in_channel"], activation=act)
    if kind == "rfb":
        return T.RFB(kw["in_channel"], kw["out_channel"], activation=act, add_sece=True)
    if kind == "asp":
        return T.ASP(kw["in_channel"], kw["out_channel"], act_fn=act, asp_rate=tuple(kw["asp_rate"]))
    return T.ResidualBlock(bias=False, BN=True, activation=act, **kw)


@both_backends
def test_golden_seg_blocks(backend):
    meta = json.load(open(os.path.join(GOLD, "seg_blocks.json")))
    G = np.load(os.path.join(GOLD, "seg_blocks.npz"))
    with BACKENDS[backend]() as dev:
        for c in meta:
            i = c["idx"]
            pre = f"blk{i}."
            m = build_block(c)
            assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == c["keys"], f"blk{i} state_dict layout"
            fill_state_dict_(m.state_dict(), seed=300 + i)
            m = m.to(dev).train()
            x = torch.from_numpy(G[pre + "x"]).to(dev).requires_grad_(True)
            y = m(x)
            assert_close(y, G[pre + "y"], TOL, f"blk{i} {c['kind']} y")
            y.backward(torch.from_numpy(G[pre + "gy"]).to(dev))
            assert_close(x.grad, G[pre + "dx"], TOL, f"blk{i} {c['kind']} dx")
            params, sd = dict(m.named_parameters()), m.state_dict()
            gmax = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith(pre + "grad."))
            for k in G.files:
                if k.startswith(pre + "grad."):
                    assert_close(params[k[len(pre) + 5:]].grad, G[k], 2e-3, k, floor=1e-3 * gmax)
                if k.startswith(pre + "buf."):
                    assert_close(sd[k[len(pre) + 4:]], G[k], TOL, k)


@both_backends
def test_golden_focal_loss(backend):
    G = np.load(os.path.join(GOLD, "focal_loss.npz"))
    with BACKENDS[backend]() as dev:
        t = torch.from_numpy(G["t"]).to(dev)
        for j in range(4):
            g, bw, ww = (float(v) for v in G[f"c{j}.cfg"])
            x = torch.from_numpy(G["x"]).to(dev).requires_grad_(True)
            loss = T.BinaryFocalLoss(g, bw, ww)(x, t)
            assert abs(loss.item() - float(G[f"c{j}.loss"])) <= 1e-5 * max(1.0, abs(float(G[f"c{j}.loss"])))
            loss.backward()
            assert_close(x.grad, G[f"c{j}.dx"], TOL, f"focal c{j} dx")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["TextSegament", "XceptionTextSegment"])
def test_golden_seg_nets_64_gpu(name):
    golden_seg_net_case(name)


def golden_seg_net_case(name, checkpoint_encoder=False):
    """A segmentation net against the fixture the REFERENCE produced (tests/golden/make_golden_seg.py): eval / train outputs, focal
    loss and every recorded gradient.  checkpoint_encoder: the same through the stage-recomputing encoder (SURVEY.md n4;
    tests/test_memory_savers.py)."""
    keys = json.load(open(os.path.join(GOLD, "seg_state_dict_keys.json")))[name]
    G = np.load(os.path.join(GOLD, name.lower() + "_64.npz"))
    with BACKENDS["gpu"]() as dev:
        m = getattr(T, name)()
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == keys
        fill_state_dict_(m.state_dict(), seed=41, gain=1.0)
        m = m.to(dev)
        if checkpoint_encoder:
            m.checkpoint_encoder = True
        x, t = torch.from_numpy(G["x"]).to(dev), torch.from_numpy(G["t"]).to(dev)
        m.eval()
        with torch.no_grad():
            ye = m(x)
        assert tuple(ye.shape) == (2, 1, 64, 64)
        assert_close(ye, G["y_eval"], TOL, name + " eval")
        assert_close(ye, G["y_eval_f64"], TOL, name + " eval vs the reference's fp64 run")
        m.train()
        y = m(x)
        assert_close(y, G["y_train"], TOL, name + " train")
        loss = T.BinaryFocalLoss(0, 1, 2)(y, t)
        assert abs(loss.item() - float(G["loss"])) < 1e-4
        loss.backward()
        params = dict(m.named_parameters())
        gmax = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith("grad."))
        n = 0
        for k in G.files:
            if k.startswith("grad."):
                # tolerance: 3e-3, or 8x the reference's own fp32-vs-fp64 discrepancy for this tensor when that is
                # larger (train-mode BN chains amplify rounding noise) -- measured against the reference's
                # fp64 gradient.  The fixture holds ONE fp32 run of the reference, i.e. one sample of that noise: on
                # entry_flow_1.2.weight (sample 6.9e-4) this package's bit-exact fp32-FMA mode lands at 1.9e-3 (2.7x) on
                # the emulator and the split-bf16 default at 3.8e-3 (5.6x) on the chip, with every kernel involved
                # exact against float64 at kernel level (tests/test_emu_kernels.py).  The 256^2 test (tests/test_parity_r2.py) measures the
                # oracle's spread over several perturbed fp32 runs instead of assuming a factor.
                ref64 = G["grad64." + k[5:]]
                scale = max(float(np.abs(G[k]).max()), 1e-3 * gmax)
                noise = float(np.abs(G[k] - ref64).max()) / scale
                assert_close(params[k[5:]].grad, ref64.astype(np.float32), max(3e-3, 8 * noise), k + " vs fp64", floor=1e-3 * gmax)
                assert_close(params[k[5:]].grad, G[k], max(3e-3, 9 * noise), k, floor=1e-3 * gmax)   # vs the fp32 run: the two noises add
                n += 1
        assert n >= 12


@both_backends
def test_golden_inpainting_loss(backend):
    """a20: InpaintingLoss (pixel + TV + perceptual + style terms, frozen MobileNetV2 feature extractor) vs the
    fixture the reference produced: loss value and gradient w.r.t. the network output."""
    G = np.load(os.path.join(GOLD, "inpainting_loss.npz"))
    keys = json.loads(str(G["keys"]))
    with BACKENDS[backend]() as dev:
        crit = T.InpaintingLoss(T.MobileNetV2(width_mult=1), feature_range=3)
        assert [[k, list(v.shape)] for k, v in crit.state_dict().items()] == keys
        fill_state_dict_(crit.state_dict(), seed=77)
        crit = crit.to(dev)
        gt, mask = torch.from_numpy(G["gt"]).to(dev), torch.from_numpy(G["mask"]).to(dev)
        out = torch.from_numpy(G["out"]).to(dev).requires_grad_(True)
        raw = torch.from_numpy(G["gt"] * G["mask"]).to(dev)
        loss = crit(raw, mask, out, gt)
        assert abs(loss.item() - float(G["loss"])) <= 1e-4 * abs(float(G["loss"]))
        loss.backward()
        assert_close(out.grad, G["dout"], TOL, "InpaintingLoss d/d(output)")


@both_backends
def test_pixel_shuffle_vs_torch(backend):
    """K12: no reference code -- oracle is stock torch.nn.PixelShuffle on CPU; exact permutation."""
    from text_segmentation_image_inpainting_amd.BaseModels import PixelShuffle
    with BACKENDS[backend]() as dev:
        for r, c in ((2, 3), (4, 1), (2, 8)):
            x = torch.randn(2, c * r * r, 5, 6)
            xo = x.clone().requires_grad_(True)
            yo = torch.nn.PixelShuffle(r)(xo)
            gy = torch.randn_like(yo)
            yo.backward(gy)
            xd = x.to(dev).requires_grad_(True)
            y = PixelShuffle(r)(xd)
            assert torch.equal(y.detach().cpu(), yo.detach())
            y.backward(gy.to(dev))
            assert torch.equal(xd.grad.cpu(), xo.grad)


@both_backends
def test_textsegament_pixel_shuffle_head_vs_stock(backend):
    """cfg 3's head variant, ``TextSegament(pixel_shuffle_head=True)``: Conv2d(128, 16, 3) -> PixelShuffle(4).  The reference
    ships no code for it, so the oracle is stock torch on the CPU in fp64: (a) the head alone, forward and
    every gradient, on a random feature map; (b) the whole network's output = stock head applied to the features the
    network itself produced (wiring), at width 0.25 so the emulator run stays short."""
    import torch.nn.functional as F
    rng = np.random.default_rng(77)
    with BACKENDS[backend]() as dev:
        m = T.TextSegament(width_mult=0.25 if backend == "emu" else 2, pixel_shuffle_head=True)
        fill_state_dict_(m.state_dict(), seed=43, gain=1.0)
        m = m.to(dev).train()
        w = m.out_conv[0].weight.detach().cpu().double().requires_grad_(True)
        b = m.out_conv[0].bias.detach().cpu().double().requires_grad_(True)
        assert tuple(w.shape) == (16, 128, 3, 3)
        # (a) the head alone
        feat = torch.from_numpy(rng.standard_normal((2, 128, 12, 10)).astype(np.float32))
        fo = feat.double().requires_grad_(True)
        yo = F.pixel_shuffle(F.conv2d(fo, w, b, padding=1), 4)
        gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
        yo.backward(gy.double())
        fd = feat.to(dev).requires_grad_(True)
        y = m.out_conv(fd)
        assert tuple(y.shape) == (2, 1, 48, 40)
        assert_close(y, yo.detach(), TOL, "pixel-shuffle head y")
        y.backward(gy.to(dev))
        assert_close(fd.grad, fo.grad, TOL, "pixel-shuffle head d/d(features)")
        assert_close(m.out_conv[0].weight.grad, w.grad, TOL, "pixel-shuffle head dW")
        assert_close(m.out_conv[0].bias.grad, b.grad, TOL, "pixel-shuffle head db")
        # (b) wiring of the whole net: x4 logits of the features it computed
        grabbed = []
        h = m.smooth_feature_4x_conv.register_forward_hook(lambda mod, inp, out: grabbed.append(out.detach().cpu().double()))
        x = torch.from_numpy(rng.standard_normal((2, 3, 32, 32)).astype(np.float32)).to(dev)
        with torch.no_grad():
            out = m(x)
        h.remove()
        assert tuple(out.shape) == (2, 1, 32, 32) and len(grabbed) == 1
        ref = F.pixel_shuffle(F.conv2d(grabbed[0], w.detach(), b.detach(), padding=1), 4)
        assert_close(out, ref, TOL, "TextSegament(pixel_shuffle_head=True) output")


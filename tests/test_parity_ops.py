"""Parity of the HIP path (through modules -> autograd -> ctypes -> C ABI -> kernels) with the
CPU oracle / the reference-generated golden fixtures.  fp32 tolerance 1e-3 max-normalised
(BASELINE.json north_star); new_mask must be bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

import text_segmentation_image_inpainting_amd as T
from oracle import pconv_oracle as O
from oracle.filler import fill_state_dict_, make_state_dict
from tests.backends import BACKENDS, both_backends
from tests.util import assert_close, assert_gradients_close, rel_err

TOL = 1e-3
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _build_op(c):
    if c["kind"] == "pconv":
        return T.PartialConv(c["cin"], c["cout"], c["k"], c["s"], c["p"], c["d"], c["groups"], c["bias"], c["same_holes"])
    if c["kind"] == "pconv1x1":
        return T.PartialConv1x1(c["cin"], c["cout"], c["k"], c["s"], c["p"], c["d"], c["groups"], c["bias"])
    return T.PartialConvNoHoles(c["cin"], c["cout"], c["k"], c["s"], c["p"], c["d"], c["groups"], c["bias"])


@both_backends
def test_golden_op_cases(backend):
    """a1/a2/a3 against the fixtures the reference produced (forward, new_mask, dx, dw, db)."""
    meta = json.load(open(os.path.join(GOLD, "pconv_ops.json")))
    G = np.load(os.path.join(GOLD, "pconv_ops.npz"))
    with BACKENDS[backend]() as dev:
        for c in meta:
            i = c["idx"]
            pre = f"op{i}."
            m = _build_op(c)
            fill_state_dict_(m.state_dict(), seed=i)
            m = m.to(dev)
            x = torch.from_numpy(G[pre + "x"]).to(dev).requires_grad_(True)
            mask = torch.from_numpy(G[pre + "mask"]).to(dev)
            y, nm = m((x, mask))
            assert tuple(y.shape) == G[pre + "y"].shape
            assert_close(y, G[pre + "y"], TOL, f"op{i} y")
            assert np.array_equal(nm.detach().cpu().numpy(), G[pre + "new_mask"]), f"op{i} new_mask not bit-exact"
            gy = torch.from_numpy(G[pre + "gy"]).to(dev)
            fin = torch.isfinite(y)
            if fin.all():
                y.backward(gy)
            else:  # NaN rows (NoHoles all-hole windows) were excluded from the reference's backward seed too
                continue
            assert_close(x.grad, G[pre + "dx"], TOL, f"op{i} dx")
            assert_close(m.feature_conv.weight.grad, G[pre + "dw"], TOL, f"op{i} dw")
            if c["bias"]:
                assert_close(m.feature_conv.bias.grad, G[pre + "db"], TOL, f"op{i} db")


@both_backends
def test_golden_pir_blocks(backend):
    """a7 PartialInvertedResidual in train mode: output, mask, input/weight grads, BN running stats."""
    meta = json.load(open(os.path.join(GOLD, "pir_blocks.json")))
    G = np.load(os.path.join(GOLD, "pir_blocks.npz"))
    with BACKENDS[backend]() as dev:
        for c in meta:
            i = c["idx"]
            pre = f"pir{i}."
            m = T.PartialInvertedResidual(c["in_c"], c["out_c"], c["k"], c["s"], c["p"], c["d"], c["t"], bias=False,
                                          BN=True, activation=torch.nn.LeakyReLU(0.3), use_1_conv=c["use_1_conv"],
                                          no_holes_1_conv=c["no_holes_1_conv"], same_holes=c["same_holes"])
            assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == c["keys"]
            fill_state_dict_(m.state_dict(), seed=100 + i)
            m = m.to(dev).train()
            x = torch.from_numpy(G[pre + "x"]).to(dev).requires_grad_(True)
            mask = torch.from_numpy(G[pre + "mask"]).to(dev)
            y, nm = m((x, mask))
            assert_close(y, G[pre + "y"], TOL, f"pir{i} y")
            assert np.array_equal(nm.detach().cpu().numpy(), G[pre + "new_mask"])
            y.backward(torch.from_numpy(G[pre + "gy"]).to(dev))
            assert_close(x.grad, G[pre + "dx"], TOL, f"pir{i} dx")
            params = dict(m.named_parameters())
            sd = m.state_dict()
            for k in G.files:
                if k.startswith(pre + "grad."):
                    assert_close(params[k[len(pre) + 5:]].grad, G[k], TOL, k)
                if k.startswith(pre + "buf."):
                    assert_close(sd[k[len(pre) + 4:]], G[k], TOL, k)


@both_backends
def test_golden_pir_blocks_all_batchnorm_fusions(backend, monkeypatch):
    """The PartialInvertedResidual fixtures again with the point-wise K6c form (BatchNorm-backward reductions in the dX
    GEMM epilogue, the default) switched off, so both forms of every fused BatchNorm entry point are held to the
    reference-generated grads."""
    from text_segmentation_image_inpainting_amd import ops
    monkeypatch.setattr(ops, "FUSE_BN_BWD_PW", False)       # the default takes the GEMM-epilogue form; here: without it
    test_golden_pir_blocks.__wrapped__(backend) if hasattr(test_golden_pir_blocks, "__wrapped__") else test_golden_pir_blocks(backend)


@both_backends
def test_golden_pir_blocks_weight_gradient_in_the_dx_pass(backend, monkeypatch):
    """K6d (the depth-wise dX + K6c pass also takes the weight gradient, the default for stride 1 / dilation 1) really runs on the
    PartialInvertedResidual fixtures -- and the same fixtures with it switched off (the separate tsii_dw_bwd_dw_bn pass)."""
    from text_segmentation_image_inpainting_amd import ops
    calls = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    run = test_golden_pir_blocks.__wrapped__ if hasattr(test_golden_pir_blocks, "__wrapped__") else test_golden_pir_blocks
    run(backend)
    # stride-1 layers: K6e, the one-pass kernel also applies the following BatchNorm's backward on load (that BatchNorm only reduces)
    assert "tsii_dw_bwd_dxdw_bn2" in calls and "tsii_bn_bwd_reduce" in calls, sorted(set(calls))
    del calls[:]
    monkeypatch.setattr(ops, "FUSE_DW_BN2_FOLD", False)
    run(backend)
    assert "tsii_dw_bwd_dxdw_bn" in calls and "tsii_dw_bwd_dxdw_bn2" not in calls and "tsii_bn_bwd_reduce" not in calls, sorted(set(calls))
    del calls[:]
    monkeypatch.setattr(ops, "FUSE_DW_DXDW", False)
    run(backend)
    assert "tsii_dw_bwd_dxdw_bn" not in calls and "tsii_dw_bwd_dw_bn" in calls, sorted(set(calls))


@both_backends
def test_deferred_batchnorm_backward_falls_back_and_refuses(backend, monkeypatch):
    """K6e's hand-over between the BatchNorm that follows a depth-wise layer and that layer's backward: (1) a switch flipped BETWEEN
    forward and backward (the one-pass kernel is no longer allowed) -> the stand-alone apply pass runs inside the layer's backward and
    the fixture gradients still hold; (2) a second consumer of the layer's raw output -> the deferral cannot be undone: RuntimeError."""
    from text_segmentation_image_inpainting_amd import ops
    meta = json.load(open(os.path.join(GOLD, "pir_blocks.json")))
    G = np.load(os.path.join(GOLD, "pir_blocks.npz"))
    c = next(c for c in meta if c["s"] == 1 and c["d"] == 1)
    i = c["idx"]
    pre = f"pir{i}."
    calls = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    with BACKENDS[backend]() as dev:
        m = T.PartialInvertedResidual(c["in_c"], c["out_c"], c["k"], c["s"], c["p"], c["d"], c["t"], bias=False, BN=True,
                                      activation=torch.nn.LeakyReLU(0.3), use_1_conv=c["use_1_conv"],
                                      no_holes_1_conv=c["no_holes_1_conv"], same_holes=c["same_holes"])
        fill_state_dict_(m.state_dict(), seed=100 + i)
        m = m.to(dev).train()
        x = torch.from_numpy(G[pre + "x"]).to(dev).requires_grad_(True)
        y, _ = m((x, torch.from_numpy(G[pre + "mask"]).to(dev)))
        monkeypatch.setattr(ops, "FUSE_DW_DXDW", False)          # after the forward decided to defer
        y.backward(torch.from_numpy(G[pre + "gy"]).to(dev))
        assert "tsii_bn_bwd_reduce" in calls and "tsii_bn_bwd_apply" in calls and "tsii_dw_bwd_dxdw_bn2" not in calls, sorted(set(calls))
        assert_close(x.grad, G[pre + "dx"], TOL, f"pir{i} dx (fallback)")
        params = dict(m.named_parameters())
        for k in G.files:
            if k.startswith(pre + "grad."):
                assert_close(params[k[len(pre) + 5:]].grad, G[k], TOL, k + " (fallback)")
        monkeypatch.setattr(ops, "FUSE_DW_DXDW", True)
        # (2) the layer's raw output consumed by its BatchNorm AND directly
        cch = 8
        g = ops.make_geom(3, 1, 1, 1)
        rng = np.random.default_rng(5)
        t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32)).to(dev)
        ones = lambda n: torch.ones(n, device=dev)
        x0 = t(1, 6, 7, cch).requires_grad_(True)
        lz = ops.bn_lazy(x0 * 1.0, ones(cch).requires_grad_(True), torch.zeros(cch, device=dev).requires_grad_(True), torch.zeros(cch, device=dev), ones(cch),
                         True, act=ops.ACT_LEAKY, slope=0.3)
        w = t(cch, 1, 3, 3).requires_grad_(True)
        yraw = ops.pconv_depthwise(lz, w, None, None, None, None, None, g)
        assert getattr(yraw, "_tsii_fold", None) is not None, "the layer offers the deferral for this geometry"
        lz2 = ops.bn_lazy(yraw, ones(cch).requires_grad_(True), torch.zeros(cch, device=dev).requires_grad_(True), torch.zeros(cch, device=dev), ones(cch),
                          True, act=ops.ACT_LEAKY, slope=0.3)
        w2 = t(4, cch, 1, 1).requires_grad_(True)
        out = ops.pconv_pointwise(lz2, w2)
        loss = out.sum() + yraw.sum()          # the second consumer
        with pytest.raises(RuntimeError, match="another consumer"):
            loss.backward()


def _imagefill_case(dev, size, batch, seed, per_channel_mask, hole_frac=0.12):
    from oracle.filler import seeded_input
    keys = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    x, mask = seeded_input(batch, 3, size, size, seed=seed, hole_frac=hole_frac, per_channel_mask=per_channel_mask)
    clean = torch.from_numpy(np.random.default_rng(seed + 2).standard_normal((batch, 3, size, size)).astype(np.float32))
    # oracle (CPU, stock torch)
    sd = make_state_dict([(k, s) for k, s in keys["ImageFill"]], seed=seed)
    for k in keys["ImageFill.trainable"]:
        sd[k].requires_grad_(True)
    yo = O.image_fill(sd, x, mask, training=True)
    lo = O.l1_mean(yo, clean)
    lo.backward()
    # product
    model = T.ImageFill()
    fill_state_dict_(model.state_dict(), seed=seed)
    model = model.to(dev).train()
    y = model((x.to(dev), mask.to(dev)))
    from text_segmentation_image_inpainting_amd import ops
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    loss = ops.l1_mean(to_nhwc(y), to_nhwc(clean.to(dev)))
    loss.backward()
    return model, sd, y, yo, loss, lo


def _check_imagefill(dev, size, batch, seed, pcm):
    model, sd, y, yo, loss, lo = _imagefill_case(dev, size, batch, seed, pcm)
    b = sd["decoder.3.0.feature_conv.bias"].detach().view(1, 3, 1, 1)
    assert_close(y.detach().cpu() - b, yo.detach() - b, TOL, "ImageFill output (bias removed, SURVEY F4)")
    assert abs(loss.item() - lo.item()) <= 1e-5 * max(1.0, abs(lo.item()))
    errs = {}
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, k
        errs[k] = rel_err(p.grad, sd[k].grad, 1e-6)
    params = dict(model.named_parameters())
    worst = assert_gradients_close(errs, 2e-3, "ImageFill grads", pairs=lambda k: (params[k].grad, sd[k].grad))      # robust to activation-kink flips, which must show as low-rank errors (tests/util.py)
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v, sd[k], TOL, k)
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(sd[k]) == 1
    return worst


def test_imagefill_train_step_emu():
    """Whole ImageFill fwd + L1 + bwd at 32x32 through the emulated kernels vs the oracle."""
    with BACKENDS["emu"]() as dev:
        _check_imagefill(dev, 32, 2, seed=3, pcm=True)


@pytest.mark.gpu
@pytest.mark.parametrize("size,batch,pcm", [(64, 2, True), (128, 2, False), (256, 1, True)])
def test_imagefill_train_step_gpu(size, batch, pcm):
    with BACKENDS["gpu"]() as dev:
        _check_imagefill(dev, size, batch, seed=size + batch, pcm=pcm)


@pytest.mark.gpu
def test_imagefill_golden_64_gpu():
    """Against the fixture the reference itself produced (train fwd, loss, sampled grads, BN buffers)."""
    keys = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    G = np.load(os.path.join(GOLD, "imagefill_64.npz"))
    with BACKENDS["gpu"]() as dev:
        model = T.ImageFill()
        assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == keys["ImageFill"]
        fill_state_dict_(model.state_dict(), seed=7)
        model = model.to(dev)
        x, mask, clean = (torch.from_numpy(G[k]).to(dev) for k in ("x", "mask", "clean"))
        b = model.decoder[3][0].feature_conv.bias.detach().view(1, 3, 1, 1).cpu()
        model.eval()
        with torch.no_grad():
            ye = model((x, mask))
        assert_close(ye.cpu() - b, torch.from_numpy(G["y_eval"]) - b, TOL, "eval output")
        model.train()
        y = model((x, mask))
        assert_close(y.detach().cpu() - b, torch.from_numpy(G["y_train"]) - b, TOL, "train output")
        assert_close(y.detach().cpu() - b, torch.from_numpy(G["y_train_f64"]).float() - b, TOL, "train output vs fp64")
        from text_segmentation_image_inpainting_amd import ops
        from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
        loss = ops.l1_mean(to_nhwc(y), to_nhwc(clean))
        assert abs(loss.item() - float(G["loss"])) < 1e-5
        loss.backward()
        params = dict(model.named_parameters())
        sd = model.state_dict()
        errs = {}
        for k in G.files:
            if k.startswith("grad."):
                errs[k] = rel_err(params[k[5:]].grad, G[k], 1e-6)
            if k.startswith("buf."):
                assert_close(sd[k[4:]], G[k], TOL, k)
        assert_gradients_close(errs, 2e-3, "ImageFill 64 fixture grads", pairs=lambda k: (params[k[5:]].grad, G[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ImageFillOrigin", "ImageFillOriginV2"])
def test_origin_models_golden_256_gpu(name):
    """a9 / a10 against the reference-generated fixtures (eval bs 1, train-mode forward bs 2) and the
    oracle's backward (train bs 2) at 256x256, the smallest size these 8-level nets accept."""
    keys = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    G = np.load(os.path.join(GOLD, name.lower() + "_256.npz"))
    x = torch.from_numpy(G["x"].astype(np.float32))
    plane = np.unpackbits(G["mask"])[: 256 * 256].reshape(1, 1, 256, 256).astype(np.float32)
    mask = torch.from_numpy(np.repeat(plane, 3, axis=1))
    fn = O.MODELS[name]
    with BACKENDS["gpu"]() as dev:
        model = getattr(T, name)()
        assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == keys[name]
        fill_state_dict_(model.state_dict(), seed=11)
        model = model.to(dev)
        b = model.decoder[7][0].feature_conv.bias.detach().view(1, 3, 1, 1).cpu()
        model.eval()
        with torch.no_grad():
            ye = model((x.to(dev), mask.to(dev)))
        assert_close(ye.cpu() - b, torch.from_numpy(G["y_eval"]) - b, TOL, name + " eval (bias removed)")
        # train mode, batch 2 (image + its flip): forward vs fixture, backward vs oracle
        x2, m2 = torch.cat([x, x.flip(3)]), torch.cat([mask, mask.flip(3)])
        clean = torch.from_numpy(np.random.default_rng(4).standard_normal((2, 3, 256, 256)).astype(np.float32))
        model.train()
        y = model((x2.to(dev), m2.to(dev)))
        assert_close(y.detach().cpu() - b, torch.from_numpy(G["y_train_b2"]) - b, TOL, name + " train b2 (bias removed)")
        from text_segmentation_image_inpainting_amd import ops
        from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
        loss = ops.l1_mean(to_nhwc(y), to_nhwc(clean.to(dev)))
        loss.backward()
        sd = make_state_dict([(k, s) for k, s in keys[name]], seed=11)
        for k in keys[name + ".trainable"]:
            sd[k].requires_grad_(True)
        yo = fn(sd, x2, m2, training=True)
        lo = O.l1_mean(yo, clean)
        lo.backward()
        assert abs(loss.item() - lo.item()) <= 1e-5 * max(1.0, abs(lo.item()))
        params = dict(model.named_parameters())
        assert_gradients_close({k: rel_err(p.grad, sd[k].grad, 1e-6) for k, p in params.items() if p.requires_grad},
                               3e-3, name + " grads", pairs=lambda k: (params[k].grad, sd[k].grad))


# cin, cout, k, s, p, d, bias, same_holes, two_plane, H
CONV_GEMM_CASES = [
    (8, 16, 3, 1, 1, 1, True, True, False, 12),      # encoder flavour (same_holes), implicit GEMM
    (16, 32, 3, 2, 1, 1, False, True, False, 13),    # stride 2, odd size
    (8, 16, 5, 2, 2, 1, False, True, False, 12),     # 5x5 s2 (ImageFillOrigin encoder)
    (12, 16, 3, 1, 2, 2, True, False, True, 10),     # dilated, decoder flavour: two mask planes, non-same-holes
    (16, 24, (1, 3), 1, (0, 1), 1, False, False, False, 9),   # RFB (1 x k)
    (40, 136, 3, 1, 1, 1, True, False, True, 9),     # > 128 output channels, K = 360 (tile tails)
    (67, 3, 3, 1, 1, 1, True, False, True, 9),       # ImageFillOrigin final layer: LDS-tiled Cout<=4 path, 16-wide tile
    (35, 3, 3, 1, 1, 1, True, False, True, 40),      # ImageFill final layer: 32-wide tile, several tiles
    (24, 4, 3, 1, 1, 1, False, False, False, 21),    # head kernels: 4 output channels, zero-padded channel groups, one plane
    (64, 2, 3, 1, 1, 1, True, True, False, 19),      # head kernels: 16-wide tile, same_holes
    (16, 32, 5, 2, 2, 1, True, True, False, 14),     # strided dX as stride phases: 5x5 s2 (9/6/6/4 taps), even size
    (24, 48, 3, 2, 1, 1, False, False, True, 15),    # 3x3 s2 (4/2/2/1 taps), odd size, two mask planes
    (128, 1, 3, 1, 1, 1, True, True, False, 11),     # 128 -> 1 logits conv: forward on the GEMM path (Cout below the gather minimum)
]


@both_backends
def test_dense_conv_implicit_gemm_vs_oracle(backend):
    """K4 on the MFMA path (gather loaders of the NT / TN GEMM kernels): forward, new_mask, dX, dW, db vs the oracle."""
    with BACKENDS[backend]() as dev:
        for idx, (cin, cout, k, s, p, d, bias, same, two, H) in enumerate(CONV_GEMM_CASES):
            m = T.PartialConv(cin, cout, k, s, p, d, 1, bias, same)
            fill_state_dict_(m.state_dict(), seed=500 + idx)
            rng = np.random.default_rng(500 + idx)
            x = torch.from_numpy(rng.standard_normal((2, cin, H, H)).astype(np.float32))
            pa = (torch.from_numpy(rng.uniform(size=(2, 1, H, H))) > 0.3).float()
            if two:
                c1 = cin // 2
                pb = (torch.from_numpy(rng.uniform(size=(2, 1, H, H))) > 0.3).float()
                mask = torch.cat([pa.expand(-1, c1, -1, -1), pb.expand(-1, cin - c1, -1, -1)], 1).contiguous()
                from text_segmentation_image_inpainting_amd.masks import MaskParts, Part
                mparts = MaskParts([Part(c1, plane=pa[:, 0].contiguous().to(dev)), Part(cin - c1, plane=pb[:, 0].contiguous().to(dev))])
            else:
                mask = pa.expand(-1, cin, -1, -1).contiguous()
                mparts = pa.expand(-1, cin, -1, -1).to(dev)          # stride-0 view -> planar fast path
            w = m.feature_conv.weight.detach().clone().requires_grad_(True)
            b = m.feature_conv.bias.detach().clone().requires_grad_(True) if bias else None
            xo = x.clone().requires_grad_(True)
            yo, nmo = O.partial_conv(xo, mask, w, b, m.feature_conv.stride, m.feature_conv.padding, m.feature_conv.dilation, 1, same)
            gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            m = m.to(dev)
            xd = x.to(dev).requires_grad_(True)
            y, nm = m((xd, mparts))
            nm_t = nm.as_tensor() if hasattr(nm, "as_tensor") else nm
            assert_close(y, yo, TOL, f"conv-gemm case {idx} y")
            assert np.array_equal(nm_t.detach().cpu().numpy(), nmo.detach().numpy()), f"case {idx} new_mask"
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, f"conv-gemm case {idx} dx")
            assert_close(m.feature_conv.weight.grad, w.grad, TOL, f"conv-gemm case {idx} dw")
            if bias:
                assert_close(m.feature_conv.bias.grad, b.grad, TOL, f"conv-gemm case {idx} db")


@both_backends
@pytest.mark.parametrize("s2d", [True, False])
def test_stem_conv_elementwise_gather_vs_oracle(backend, s2d, monkeypatch):
    """3-channel stems on the implicit-GEMM path -- as a space-to-depth stride-1 conv on the vector gather (K4b) and on
    the element-wise gather: per-channel mask (ImageFill stem), same_holes plane mask (ImageFillOrigin stem) and a plain
    3x3 s2 conv (MobileNetV2 / Xception first layer)."""
    from text_segmentation_image_inpainting_amd import ops
    from text_segmentation_image_inpainting_amd.BaseModels import Conv2d
    monkeypatch.setattr(ops, "USE_STEM_S2D", s2d)
    with BACKENDS[backend]() as dev:
        for idx, (cout, k, s, p, bias, same, pcm) in enumerate([(16, 7, 2, 3, True, False, True), (32, 7, 2, 3, True, True, False),
                                                              (24, 5, 2, 2, False, True, False)]):
            rng = np.random.default_rng(700 + idx)
            H = 18
            m = T.PartialConv(3, cout, k, s, p, 1, 1, bias, same)
            fill_state_dict_(m.state_dict(), seed=700 + idx)
            x = torch.from_numpy(rng.standard_normal((2, 3, H, H)).astype(np.float32))
            if pcm:
                mask = (torch.from_numpy(rng.uniform(size=(2, 3, H, H))) > 0.35).float()
                mask[:, :, 4:9, 5:11] = 0
            else:
                mask = (torch.from_numpy(rng.uniform(size=(2, 1, H, H))) > 0.35).float().expand(-1, 3, -1, -1).contiguous()
            w = m.feature_conv.weight.detach().clone().requires_grad_(True)
            b = m.feature_conv.bias.detach().clone().requires_grad_(True) if bias else None
            yo, nmo = O.partial_conv(x, mask, w, b, s, p, 1, 1, same)
            gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            m = m.to(dev)
            y, nm = m((x.to(dev), mask.to(dev)))
            assert_close(y, yo, TOL, f"stem {idx} y")
            assert np.array_equal(nm.detach().cpu().numpy(), nmo.detach().numpy()), f"stem {idx} new_mask"
            y.backward(gy.to(dev))
            assert_close(m.feature_conv.weight.grad, w.grad, TOL, f"stem {idx} dw")
            if bias:
                assert_close(m.feature_conv.bias.grad, b.grad, TOL, f"stem {idx} db")
        # plain conv (no mask): MobileNetV2 first layer 3 -> 32, 3x3 s2
        conv = Conv2d(3, 32, 3, 2, 1, bias=False)
        fill_state_dict_(conv.state_dict(), seed=710)
        x = torch.from_numpy(np.random.default_rng(710).standard_normal((2, 3, 17, 17)).astype(np.float32))
        w = conv.weight.detach().clone().requires_grad_(True)
        yo = torch.nn.functional.conv2d(x, w, None, 2, 1)
        gy = torch.from_numpy(np.random.default_rng(711).standard_normal(tuple(yo.shape)).astype(np.float32))
        yo.backward(gy)
        conv = conv.to(dev)
        y = conv(x.to(dev))
        assert_close(y, yo, TOL, "plain stem y")
        y.backward(gy.to(dev))
        assert_close(conv.weight.grad, w.grad, TOL, "plain stem dw")


@pytest.mark.gpu
def test_train_step_graph_replay_matches_eager_gpu():
    """The HIP-graph replay of the training step (train_step.FlatSGDTrainer.capture) does exactly the work of the
    Python-launched step: same loss and bit-identical parameters / BatchNorm buffers after two steps."""
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    with BACKENDS["gpu"]() as dev:
        corrupted, mask, clean = make_batch(2, 64, seed0=5)
        corrupted, mask = corrupted.to(dev), mask.to(dev)
        clean_nhwc = to_nhwc(clean.to(dev))
        results = []
        for graphed in (False, True):
            torch.manual_seed(0)
            model = T.ImageFill()
            fill_state_dict_(model.state_dict(), seed=11)
            model = model.to(dev).train()
            tr = FlatSGDTrainer(model, lr=1e-2, momentum=0.9, weight_decay=1e-4)
            if graphed:
                tr.capture(corrupted, mask, clean_nhwc)
                losses = [float(tr.step_graph().item()) for _ in range(2)]
            else:
                losses = [float(tr.step(corrupted, mask, clean_nhwc).item()) for _ in range(2)]
            results.append((losses, tr.flat_param.clone(), {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k}))
        (l0, p0, b0), (l1, p1, b1) = results
        assert l0 == l1, (l0, l1)
        assert torch.equal(p0, p1)
        for k in b0:
            assert torch.equal(b0[k], b1[k]), k


@both_backends
def test_depthwise_strip_kernels_vs_oracle(backend):
    """K2 marching-strip stencils (stride 1 d 1/2, stride 2) incl. tile tails, several strip chunks and channel tails:
    forward, new_mask, dX, dW, db of the depth-wise PartialConv vs the oracle."""
    with BACKENDS[backend]() as dev:
        for idx, (c, s, d, H, W) in enumerate([(40, 1, 1, 21, 37), (36, 1, 2, 19, 18), (40, 2, 1, 22, 37), (64, 2, 1, 33, 16),
                                               (36, 1, 4, 20, 23), (32, 1, 8, 33, 18)]):
            m = T.PartialConv(c, c, 3, s, d, d, c, True, True)
            fill_state_dict_(m.state_dict(), seed=900 + idx)
            rng = np.random.default_rng(900 + idx)
            x = torch.from_numpy(rng.standard_normal((2, c, H, W)).astype(np.float32))
            pa = (torch.from_numpy(rng.uniform(size=(2, 1, H, W))) > 0.3).float()
            pa[:, :, 3:9, 2:8] = 0
            mask = pa.expand(-1, c, -1, -1).contiguous()
            w = m.feature_conv.weight.detach().clone().requires_grad_(True)
            b = m.feature_conv.bias.detach().clone().requires_grad_(True)
            xo = x.clone().requires_grad_(True)
            yo, nmo = O.partial_conv(xo, mask, w, b, s, d, d, c, True)
            gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            m = m.to(dev)
            xd = x.to(dev).requires_grad_(True)
            y, nm = m((xd, pa.expand(-1, c, -1, -1).to(dev)))
            assert_close(y, yo, TOL, f"dw strip case {idx} y")
            assert np.array_equal(nm.detach().cpu().numpy(), nmo.detach().numpy()), f"case {idx} new_mask"
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, f"dw strip case {idx} dx")
            assert_close(m.feature_conv.weight.grad, w.grad, TOL, f"dw strip case {idx} dw")
            assert_close(m.feature_conv.bias.grad, b.grad, TOL, f"dw strip case {idx} db")


@both_backends
def test_dense_block_statistics_from_gemm_epilogue(backend, monkeypatch):
    """conv(3x3 dense, implicit GEMM) + BatchNorm block: the statistics taken from the GEMM epilogue partials (K6b) give the
    same outputs, running statistics and gradients as the separate statistics pass."""
    from text_segmentation_image_inpainting_amd import partial_convolution as pc
    with BACKENDS[backend]() as dev:
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setattr(pc, "FUSE_BN", mode)
            torch.manual_seed(0)
            blk = pc.partial_convolution_block(24, 40, 3, 1, 1, 1, BN=True, activation=torch.nn.LeakyReLU(0.3), same_holes=True)
            fill_state_dict_(blk.state_dict(), seed=31)
            blk = blk.to(dev).train()
            rng = np.random.default_rng(31)
            x = torch.from_numpy(rng.standard_normal((2, 24, 19, 23)).astype(np.float32)).to(dev).requires_grad_(True)
            m = (torch.from_numpy(rng.uniform(size=(2, 1, 19, 23))) > 0.3).float().expand(-1, 24, -1, -1).to(dev)
            from text_segmentation_image_inpainting_amd.BaseModels import run_nhwc, to_nhwc
            from text_segmentation_image_inpainting_amd.masks import as_parts
            y, mp = run_nhwc(blk, to_nhwc(x), as_parts(m))
            y.square().sum().backward()
            res[mode] = (y.detach().cpu(), x.grad.cpu(), blk[0].feature_conv.weight.grad.cpu(), blk[1].bn_act[0].weight.grad.cpu(),
                         blk[1].bn_act[0].running_var.cpu().clone())
        for a, b, what in zip(res["1"], res["0"], ("y", "dx", "dw", "dgamma", "running_var")):
            assert_close(a, b, 2e-5, "dense block " + what, floor=1e-6)


@pytest.mark.gpu
def test_imagefill_full_size_properties_gpu():
    """BASELINE size (512x512) checks that need no oracle run: the forward is deterministic (bit-identical repeats), an
    image's eval-mode output does not depend on its batch mates, new_mask planes stay binary and nested (holes only
    shrink layer by layer), and two identical training steps produce bit-identical gradients."""
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    with BACKENDS["gpu"]() as dev:
        torch.manual_seed(0)
        model = T.ImageFill()
        fill_state_dict_(model.state_dict(), seed=3)
        model = model.to(dev)
        corrupted, mask, clean = make_batch(4, 512, seed0=77)
        corrupted, mask, clean = corrupted.to(dev), mask.to(dev), clean.to(dev)
        model.eval()
        with torch.no_grad():
            y1 = model((corrupted, mask))
            y2 = model((corrupted, mask))
            assert torch.equal(y1, y2)
            assert torch.isfinite(y1).all()
            for i in (0, 3):
                yi = model((corrupted[i:i + 1], mask[i:i + 1]))
                assert_close(yi, y1[i:i + 1], 1e-6, f"batch independence, image {i}")
            # mask path alone: a partial conv's new mask is binary and contains the old valid region
            from text_segmentation_image_inpainting_amd.masks import as_parts
            stem = model.encoder[0][0]
            _, mp1 = stem.forward_nhwc(to_nhwc(corrupted), as_parts(mask))
            nm = mp1.as_tensor()
            assert set(torch.unique(nm).tolist()) <= {0.0, 1.0}
            pooled = torch.nn.functional.max_pool2d(mask[:, :1], 2)     # valid at stride 2 wherever any input pixel was valid
            assert bool((nm[:, :1] >= pooled).all())
        grads = []
        for _ in range(2):
            torch.manual_seed(0)
            m2 = T.ImageFill()
            fill_state_dict_(m2.state_dict(), seed=3)
            tr = FlatSGDTrainer(m2.to(dev).train(), lr=1e-3)
            tr.forward_backward(corrupted, mask, to_nhwc(clean))
            tr.reduce_gradients()
            grads.append(tr.flat_grad.clone())
        assert torch.equal(grads[0], grads[1])


@both_backends
@pytest.mark.parametrize("n,h,w,c1,c2,cout,premult,low_mask", [(2, 12, 40, 32, 3, 3, False, False), (1, 18, 34, 24, 4, 4, True, False),
                                                                (2, 16, 16, 64, 3, 3, True, False), (1, 8, 66, 32, 3, 2, False, False),
                                                                (2, 16, 64, 32, 3, 3, False, True), (2, 32, 64, 64, 3, 3, True, True),
                                                                (1, 16, 128, 32, 3, 2, False, True), (1, 48, 192, 32, 3, 3, False, True),
                                                                (1, 16, 64, 32, 3, 3, True, True), (1, 32, 128, 64, 4, 1, False, True),
                                                                (2, 12, 40, 32, 3, 3, False, True), (1, 18, 34, 24, 4, 4, True, True)])
def test_head_over_virtual_concat(backend, n, h, w, c1, c2, cout, premult, low_mask):
    """K4c: the 3x3 output head over cat(nearest-x2(low), skip) WITHOUT the concatenated tensor (tsii_head_cat_*) against the
    same head over the materialised concatenation (tsii_upcat_fwd + tsii_dense_*): output, d low, d skip, dW, dbias; two mask
    planes with holes (or a pre-multiplied skip part, as at ImageFill's input level), maps that are not tile multiples.
    ``low_mask``: the low part's plane is the nearest-x2 up-sampling of the low tensor's own plane, which the VirtualCat carries
    (what the decoder hands over) -- the weight gradient then runs on the matrix cores over low-resolution pixels (K4d,
    tsii_head_cat_bwd_dw_low) where that kernel exists (c1 = 32 / 64, cout <= 3, whole 16 x 64 tiles); the emulator run caps its grid at 3 blocks so
    that blocks walk several tiles."""
    from text_segmentation_image_inpainting_amd import _lib, ops
    rng = np.random.default_rng(n * 1000 + h * 10 + w + c1)
    calls = []
    real = ops.call
    with BACKENDS[backend]() as dev:
        low = torch.from_numpy(rng.standard_normal((n, h // 2, w // 2, c1)).astype(np.float32)).to(dev)
        skip = torch.from_numpy(rng.standard_normal((n, h, w, c2)).astype(np.float32)).to(dev)
        wt = torch.from_numpy((rng.standard_normal((cout, c1 + c2, 3, 3)) * 0.2).astype(np.float32)).to(dev)
        bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
        r0_low = None
        if low_mask:
            r0_low = torch.from_numpy((rng.uniform(size=(n, h // 2, w // 2)) > 0.3).astype(np.float32)).to(dev)
            r0 = r0_low.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
        else:
            r0 = torch.from_numpy((rng.uniform(size=(n, h, w)) > 0.3).astype(np.float32)).to(dev)
        r1 = None if premult else torch.from_numpy((rng.uniform(size=(n, h, w)) > 0.3).astype(np.float32)).to(dev)
        g = ops.make_geom((3, 3), (1, 1), (1, 1), (1, 1))
        p1 = r1 if r1 is not None else torch.ones_like(r0)
        denom, new_mask, inv = ops.mask_update(r0, float(c1), p1, float(c2), g, 1.0, True)
        gy = torch.from_numpy(rng.standard_normal((n, h, w, cout)).astype(np.float32)).to(dev)
        res = []
        hook = _lib.lib().tsii_emu_set_head_blocks if backend == "emu" else None      # test-only symbol of the emulator build
        ops.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        if hook is not None:
            hook(3)
        try:
            for fused in (True, False):
                a, b = low.clone().requires_grad_(True), skip.clone().requires_grad_(True)
                ww, bb = wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
                if fused:
                    vc = ops.VirtualCat(a, b, r0_low)
                    assert ops.head_cat_ok(vc, cout, g)
                    y = ops.pconv_head_cat(vc, ww, bb, r0, r1, denom, new_mask, inv)
                else:
                    y = ops.pconv_dense(ops.upcat(a, b), ww, bb, None, r0, c1, r1, denom, new_mask, inv, g)
                y.backward(gy)
                res.append((y.detach(), a.grad, b.grad, ww.grad, bb.grad))
        finally:
            ops.call = real
            if hook is not None:
                hook(0)
        for name, u, v in zip(("y", "d low", "d skip", "dW", "dbias"), *res):
            assert_close(u, v, 2e-6, f"head over virtual concat: {name}", floor=1e-6)
    matrix_core = low_mask and c1 in (32, 64) and cout <= 3 and h % 16 == 0 and w % 64 == 0      # whole 16 x 64 tiles
    if matrix_core and c1 == 32:
        # the skip as a data tensor (no gradient): weight gradient and d low come from ONE kernel (tsii_head_cat_bwd_low)
        calls2 = []
        with BACKENDS[backend]() as dev2:
            ops.call = lambda name, *a: (calls2.append(name), real(name, *a))[1]
            try:
                a2 = low.to(dev2).clone().requires_grad_(True)
                ww2, bb2 = wt.to(dev2).clone().requires_grad_(True), bias.to(dev2).clone().requires_grad_(True)
                y2 = ops.pconv_head_cat(ops.VirtualCat(a2, skip.to(dev2), r0_low.to(dev2)), ww2, bb2, r0.to(dev2), None if r1 is None else r1.to(dev2),
                                        denom.to(dev2), new_mask.to(dev2), inv.to(dev2))
                y2.backward(gy.to(dev2))
            finally:
                ops.call = real
            assert "tsii_head_cat_bwd_low" in calls2 and "tsii_head_cat_bwd_dx" not in calls2, sorted(set(calls2))
            assert_close(a2.grad, res[1][1], 2e-6, "fused d low", floor=1e-6)
            assert_close(ww2.grad, res[1][3], 2e-6, "dW beside the fused d low", floor=1e-6)
            assert_close(bb2.grad, res[1][4], 2e-6, "dbias beside the fused d low", floor=1e-6)
    assert ("tsii_head_cat_bwd_dw_low" in calls) == matrix_core and ("tsii_head_cat_bwd_dw" in calls) == (not matrix_core), sorted(set(calls))
    assert ("tsii_head_cat_fwd_low" in calls) == (matrix_core and c2 == 3) and ("tsii_head_cat_fwd" in calls) == (not (matrix_core and c2 == 3))

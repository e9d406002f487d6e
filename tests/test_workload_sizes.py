"""The BASELINE.json configurations AT THEIR WORKLOAD SIZE (GPU only): properties that need no oracle run -- determinism, an
image's independence of its batch mates, bit-exact masks, the fused forms against their unfused spelling, finite and repeatable
gradients.  (Numerical parity with the reference is pinned at the sizes the fixtures / the CPU oracle reach: test_parity_*.)

  cfg 2  ImageFill 512x512, batch 32 (the bench step)             test_imagefill_bs32_*
  cfg 3  TextSegament(pixel_shuffle_head=True) 512x512, batch 8   test_textsegament_pixel_shuffle_512_*   (bench batch: 64)
  cfg 5  XceptionTextSegment 1024x1024, batch 2, products 1 and 6 test_xception_1024_*                    (bench batch: 8)
"""
import numpy as np
import pytest
import torch

import text_segmentation_image_inpainting_amd as T
from oracle.filler import fill_state_dict_
from tests.backends import BACKENDS
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _imagefill(dev, seed=3):
    torch.manual_seed(0)
    m = T.ImageFill()
    fill_state_dict_(m.state_dict(), seed=seed)
    return m.to(dev)


def test_imagefill_bs32_batch_independence_and_masks_gpu():
    """cfg 2 at batch 32, eval mode: every checked image equals its batch-of-one run (<= 1e-6 of the output range), the mask
    pyramid of the encoder is BIT-exact between the two runs, and the whole forward is bit-identical when repeated."""
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.masks import as_parts
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    with BACKENDS["gpu"]() as dev:
        model = _imagefill(dev).eval()
        corrupted, mask, _ = make_batch(32, 512, seed0=500)
        corrupted, mask = corrupted.to(dev), mask.to(dev)
        with torch.no_grad():
            y = model((corrupted, mask))
            assert tuple(y.shape) == (32, 3, 512, 512) and bool(torch.isfinite(y).all())
            assert torch.equal(y, model((corrupted, mask)))
            _, _, _, fm = model._encode(to_nhwc(corrupted), as_parts(mask))
            pyramid = [m.as_tensor()[:, :1].clone() for m in fm]
            for i in (0, 13, 31):
                yi = model((corrupted[i:i + 1], mask[i:i + 1]))
                assert_close(yi, y[i:i + 1], 1e-6, f"ImageFill 512^2: image {i} of 32 vs alone")
                _, _, _, fmi = model._encode(to_nhwc(corrupted[i:i + 1]), as_parts(mask[i:i + 1]))
                for lvl, (a, b) in enumerate(zip(fmi, pyramid)):
                    assert torch.equal(a.as_tensor()[:, :1], b[i:i + 1]), f"mask level {lvl}, image {i}"


@pytest.mark.parametrize("n,h,wd,c,stride,masked", [(8, 256, 256, 384, 1, True), (8, 256, 256, 256, 2, True), (16, 128, 128, 768, 1, False)])
def test_depthwise_one_pass_backward_at_the_headline_shapes_gpu(n, h, wd, c, stride, masked):
    """K6d / K6e at ImageFill's first-level layer shapes (a quarter of the bench batch) through the C ABI on the chip: the one-pass
    entry points against the passes they replace -- dX and the K6c partial rows bit for bit, the weight gradient to summation order,
    the folded BatchNorm backward to rounding.  Random hole masks (the plane the 3x3 count leaves)."""
    from text_segmentation_image_inpainting_amd import _lib, ops
    from text_segmentation_image_inpainting_amd.ops import call, ptr
    with BACKENDS["gpu"]() as dev:
        L = _lib.lib()
        g = ops.make_geom(3, stride, 1, 1)
        ho, wo = g.out_hw(h, wd)
        gen = torch.Generator(device="cpu").manual_seed(n * 1000 + c + stride)
        rnd = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
        y1 = rnd(n, h, wd, c) * 1.3 + 0.2                       # raw input of the layer's input BatchNorm
        w = rnd(c, 1, 3, 3)
        rmask = inv = None
        if masked:
            rmask = (torch.rand(n, h, wd, generator=gen) > 0.15).float().to(dev)
            denom, new_mask, inv = ops.mask_update(rmask, 1.0, None, 0.0, g, float(c), True)
        mean1, var1 = rnd(c) * 0.2, torch.rand(c, generator=gen).to(dev) + 0.5
        gam1, bet1 = torch.rand(c, generator=gen).to(dev) + 0.5, rnd(c) * 0.3
        st = _lib.stream()
        ws = torch.empty(9 * c + 16, device=dev)
        rows = int(L.tsii_dw_bwd_stat_rows(n, h, wd, c, *g))
        dwb = int(L.tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, *g))
        assert rows > 0 and dwb == 4 * rows * 9 * c
        dy = rnd(n, ho, wo, c)
        # the separate passes
        dx0, part0 = torch.empty(n, h, wd, c, device=dev), torch.empty(rows, 2, c, device=dev)
        call("tsii_dw_bwd_dx_bn", ptr(dy), ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo, ptr(y1), ptr(mean1), ptr(var1), ptr(gam1), ptr(bet1),
             1e-5, 2, 0.3, ptr(dx0), ptr(part0), ptr(ws), st)
        sc1 = gam1 / torch.sqrt(var1 + 1e-5); sh1 = bet1 - mean1 * sc1
        nb = int(L.tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, 3, 3))
        wsw = torch.empty(nb // 4 + 4, device=dev)
        dw0 = torch.empty_like(w)
        keep = new_mask if masked else None
        call("tsii_dw_bwd_dw_bn", ptr(dy), ptr(inv), ptr(keep), ptr(y1), ptr(rmask), n, h, wd, c, *g, ho, wo, ptr(sc1), ptr(sh1), 2, 0.3, ptr(dw0), None,
             ptr(wsw), nb, st)
        # K6d
        dx1, part1, dw1 = torch.full_like(dx0, float("nan")), torch.full_like(part0, float("nan")), torch.full_like(dw0, float("nan"))
        wsd = torch.empty(dwb // 4, device=dev)
        call("tsii_dw_bwd_dxdw_bn", ptr(dy), ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo, ptr(y1), ptr(mean1), ptr(var1), ptr(gam1), ptr(bet1),
             1e-5, 2, 0.3, ptr(dx1), ptr(part1), ptr(dw1), ptr(ws), ptr(wsd), dwb, st)
        assert torch.equal(dx1, dx0) and torch.equal(part1, part0)
        scale = float(dw0.abs().max())
        assert float((dw1 - dw0).abs().max()) <= 2e-5 * scale, (float((dw1 - dw0).abs().max()), scale)
        # K6e: the same with dy = BatchNorm2-backward(da2, y2) applied on load (stride 1: the lean strip kernel; stride 2: the parity strips)
        assert int(L.tsii_dw_bwd_dxdw_fold_ok(n, h, wd, c, *g)) == 1
        m2 = n * ho * wo
        da2, y2 = rnd(n, ho, wo, c), rnd(n, ho, wo, c) * 1.5 + 0.1
        mean2, var2 = rnd(c) * 0.2, torch.rand(c, generator=gen).to(dev) + 0.5
        gam2, bet2 = torch.rand(c, generator=gen).to(dev) + 0.5, rnd(c) * 0.3
        xh2 = (y2 - mean2) / torch.sqrt(var2 + 1e-5)
        dz2 = da2 * torch.where(xh2 * gam2 + bet2 > 0, 1.0, 0.3)
        part2 = torch.stack([dz2.reshape(-1, c).sum(0), (dz2 * xh2).reshape(-1, c).sum(0)]).reshape(1, 2, c).contiguous()
        coef = torch.empty(6, c, device=dev); dg2, db2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        rb = int(L.tsii_bn_bwd_reduce_ws_bytes(1, c))
        wsr = torch.empty(rb // 4 + 4, device=dev)
        call("tsii_bn_bwd_reduce", ptr(mean2), ptr(var2), ptr(gam2), ptr(bet2), 1e-5, 1, ptr(part2), 1, m2, c, ptr(dg2), ptr(db2), ptr(coef), ptr(wsr), rb, st)
        dy2 = torch.empty_like(da2)
        call("tsii_bn_bwd_apply", ptr(da2), ptr(y2), m2, c, ptr(coef), 2, 0.3, ptr(dy2), st)
        dxA, partA, dwA = torch.empty_like(dx0), torch.empty_like(part0), torch.empty_like(dw0)
        call("tsii_dw_bwd_dxdw_bn", ptr(dy2), ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo, ptr(y1), ptr(mean1), ptr(var1), ptr(gam1), ptr(bet1),
             1e-5, 2, 0.3, ptr(dxA), ptr(partA), ptr(dwA), ptr(ws), ptr(wsd), dwb, st)
        dxB, partB, dwB = torch.full_like(dx0, float("nan")), torch.full_like(part0, float("nan")), torch.full_like(dw0, float("nan"))
        call("tsii_dw_bwd_dxdw_bn2", ptr(da2), ptr(y2), ptr(coef), 2, 0.3, ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo,
             ptr(y1), ptr(mean1), ptr(var1), ptr(gam1), ptr(bet1), 1e-5, 2, 0.3, ptr(dxB), ptr(partB), ptr(dwB), ptr(ws), ptr(wsd), dwb, st)
        assert bool(torch.isfinite(dxB).all()) and bool(torch.isfinite(partB).all())
        assert float((dxB - dxA).abs().max()) <= 4e-6 * max(1.0, float(dxA.abs().max()))
        assert float((dwB - dwA).abs().max()) <= 2e-5 * float(dwA.abs().max())
        pa, pb = partA.sum(0), partB.sum(0)
        assert float((pb - pa).abs().max()) <= 1e-4 * float(dxA.abs().reshape(-1, c).sum(0).max())


def test_imagefill_bs32_split_resolution_decoder_equals_concat_gpu(monkeypatch):
    """K7b at the bench size: the decoder's 1x1 expand convolutions with their low half at low resolution (no concatenated tensor)
    against the same network with the concatenation materialised (K7 + one product), train mode, batch 32: output, loss, running
    statistics and every gradient.  Two fp32 evaluation orders of the same sums: 1e-5 of each tensor's range on the output,
    5e-4 on gradients (a LeakyReLU kink may flip: tests/util.py)."""
    from text_segmentation_image_inpainting_amd import partial_convolution as pc
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    with BACKENDS["gpu"]() as dev:
        corrupted, mask, clean = make_batch(32, 512, seed0=900)
        corrupted, mask, clean = corrupted.to(dev), mask.to(dev), to_nhwc(clean.to(dev))
        out = {}
        for fused in (True, False):
            monkeypatch.setattr(pc, "FUSE_UPCAT", fused)
            model = _imagefill(dev, seed=5).train()
            tr = FlatSGDTrainer(model, lr=1e-3)
            loss = tr.forward_backward(corrupted, mask, clean)
            tr.reduce_gradients()
            with torch.no_grad():
                y = model.eval()((corrupted[:2], mask[:2]))
            out[fused] = (float(loss), tr.flat_grad.clone(), y, {k: v.clone() for k, v in model.state_dict().items() if "running" in k},
                          {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
            tr.close()
            del tr, model
        (l1, g1, y1, rs1, pg1), (l0, g0, y0, rs0, pg0) = out[True], out[False]
        assert abs(l1 - l0) <= 1e-6 * abs(l0)
        assert_close(y1, y0, 1e-5, "eval output after the step's statistics update")
        for k in rs0:
            assert_close(rs1[k], rs0[k], 1e-5, f"running statistic {k}")
        worst = max((float((pg1[k] - pg0[k]).abs().max() / pg0[k].abs().max().clamp_min(1e-12)), k) for k in pg0)
        assert worst[0] <= 5e-4, worst


def test_textsegament_pixel_shuffle_512_properties_gpu():
    """cfg 3 with ITS head (Conv2d(128, 16, 3) -> PixelShuffle(4)) at 512x512, batch 8: deterministic forward, batch
    independence, output = stock torch head applied to the features the net itself produced (wiring at size), finite gradients
    for every trainable parameter, bit-identical when the step is repeated."""
    import torch.nn.functional as F
    from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
    with BACKENDS["gpu"]() as dev:
        x, t = make_seg_batch(8, 512, seed0=310)
        x, t = x.to(dev), t.to(dev)

        def build():
            torch.manual_seed(0)
            m = T.TextSegament(pixel_shuffle_head=True)
            fill_state_dict_(m.state_dict(), seed=48, gain=1.0)
            return m.to(dev)
        m = build().eval()
        grabbed = []
        h = m.smooth_feature_4x_conv.register_forward_hook(lambda mod, inp, out: grabbed.append(out.detach()))
        with torch.no_grad():
            y1, y2 = m(x), m(x)
            h.remove()
            assert tuple(y1.shape) == (8, 1, 512, 512) and torch.equal(y1, y2) and bool(torch.isfinite(y1).all())
            for i in (0, 6):
                assert_close(m(x[i:i + 1]), y1[i:i + 1], 1e-5, f"batch independence, image {i}")
            w, b = m.out_conv[0].weight.detach().double(), m.out_conv[0].bias.detach().double()
            ref = F.pixel_shuffle(F.conv2d(grabbed[0].double(), w, b, padding=1), 4)      # stock torch head, fp64, same device
            assert tuple(grabbed[0].shape) == (8, 128, 128, 128)
            assert_close(y1, ref, 1e-4, "pixel-shuffle head at 512^2 vs the stock head on the net's own features")
        crit = T.BinaryFocalLoss(0, 1, 2)
        grads = []
        for _ in range(2):
            m2 = build().train()
            loss = crit(m2(x), t)
            loss.backward()
            g = [p.grad for p in m2.parameters() if p.requires_grad]
            assert bool(torch.isfinite(loss)) and all(v is not None and bool(torch.isfinite(v).all()) for v in g)
            grads.append(torch.cat([v.reshape(-1) for v in g]))
            del m2
        assert torch.equal(grads[0], grads[1])


def test_textsegament_cfg3_bs64_checkpointed_step_gpu():
    """cfg 3 at ITS batch (TextSegament + pixel-shuffle head, 512x512, 64 images; `bench.py --model TextSegament --batch 64
    --pixel-shuffle --checkpoint`): the step with the encoder stages recomputed in backward (MobileNetV2.forward_checkpoint,
    /root/reference/models/MobileNetV2.py:109-111) gives the loss, every gradient and every BatchNorm buffer of the plain step
    BIT FOR BIT (the recomputation pass leaves running statistics alone), with less than 40 % of its peak memory;
    everything finite; batch independence of the eval forward at that batch."""
    from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
    with BACKENDS["gpu"]() as dev:
        x, t = make_seg_batch(64, 512, seed0=311)
        x, t = x.to(dev), t.to(dev)
        crit = T.BinaryFocalLoss(0, 1, 2)
        res = []
        for ckpt in (True, False):
            torch.manual_seed(0)
            m = T.TextSegament(pixel_shuffle_head=True)
            fill_state_dict_(m.state_dict(), seed=49, gain=1.0)
            m = m.to(dev).train()
            m.checkpoint_encoder = ckpt
            torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            loss = crit(m(x), t)
            loss.backward()
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated() - base
            g = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}
            assert bool(torch.isfinite(loss)) and all(bool(torch.isfinite(v).all()) for v in g.values())
            bufs = {k: v.clone() for k, v in m.named_buffers()}
            res.append((loss.detach().clone(), g, bufs, peak))
            if ckpt:
                m.eval()
                with torch.no_grad():
                    y = m(x)
                    for i in (0, 37, 63):
                        assert_close(m(x[i:i + 1]), y[i:i + 1], 1e-5, f"batch independence at bs 64, image {i}")
                del y
            del m, loss
        (l1, g1, b1, p1), (l0, g0, b0, p0) = res
        assert torch.equal(l1, l0)
        assert len(g1) == len(g0) >= 100
        for k in g0:
            assert torch.equal(g1[k], g0[k]), k
        for k in b0:
            assert torch.equal(b1[k], b0[k]), k
        assert p1 < 0.4 * p0, (p1 / 2**30, p0 / 2**30)


@pytest.mark.parametrize("products", [6, 1])
def test_xception_1024_properties_gpu(products, capsys):
    """cfg 5's network at its size (XceptionTextSegment, 1024x1024; batch 2 here, 8 in the bench) in the fp32-class arithmetic
    (6 partial products) and in the config's "mixed bf16" one (products = 1: bf16 operands in every matrix product): deterministic
    forward, batch independence, finite gradients for every trainable parameter, bit-identical repeats; and the bf16-operand
    eval output within 5e-2 (of the output range) of the 6-product one."""
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
    with BACKENDS["gpu"]() as dev:
        saved = _lib._GEMM_PRODUCTS
        try:
            x, t = make_seg_batch(2, 1024, seed0=320)
            x, t = x.to(dev), t.to(dev)

            def build():
                torch.manual_seed(0)
                m = T.XceptionTextSegment()
                fill_state_dict_(m.state_dict(), seed=49, gain=1.0)
                return m.to(dev)
            m = build().eval()
            with torch.no_grad():
                _lib.set_gemm_products(6)
                y6 = m(x)
                _lib.set_gemm_products(products)
                y1, y2 = m(x), m(x)
                assert tuple(y1.shape) == (2, 1, 1024, 1024) and torch.equal(y1, y2) and bool(torch.isfinite(y1).all())
                assert_close(m(x[1:2]), y1[1:2], 1e-5 if products == 6 else 1e-4, "batch independence, image 1")
                e = float((y1 - y6).abs().max() / y6.abs().max())
                if products == 1:
                    assert e <= 5e-2, e
                else:
                    assert e == 0.0
            crit = T.BinaryFocalLoss(0, 1, 2)
            grads = []
            for _ in range(2):
                m2 = build().train()
                loss = crit(m2(x), t)
                loss.backward()
                g = [p.grad for p in m2.parameters() if p.requires_grad]
                assert bool(torch.isfinite(loss)) and all(v is not None and bool(torch.isfinite(v).all()) for v in g)
                grads.append(torch.cat([v.reshape(-1) for v in g]))
                del m2
            assert torch.equal(grads[0], grads[1])
        finally:
            _lib.set_gemm_products(saved)
        with capsys.disabled():
            print(f"\n[cfg 5 at size] XceptionTextSegment 1024^2 b2, products={products}: eval output vs the 6-product run {e:.2e}")

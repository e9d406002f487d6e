"""bf16 ACTIVATION STORAGE (BASELINE config 5) at module level: the segmentation building blocks and XceptionTextSegment with
activations / activation gradients stored as bf16 (ops.set_activation_storage) against (a) the SAME modules in fp32 storage and
(b) the reference-generated fixtures / the fp64 oracle.  What is asserted, and why the bars are what they are:

* One stored tensor carries a relative rounding error of at most 2^-9 (RNE).  A block of L stored tensors in eval-mode arithmetic
  (no batch statistics) lands within ~sqrt(L) * 2^-9 of the fp32 result: bars of 2e-2 (outputs) / 5e-2 (gradients) of the
  tensor's largest entry hold for EVERY tensor.
* Train-mode BatchNorm chains amplify operand rounding (~1e4x through ~100 layers); at block level (6 convs) the
  same bars still hold for every tensor, at net level they hold for the outputs, the loss and the decoder, while the encoder's
  gradients are bounded RELATIVE to what bf16-rounded OPERANDS alone do to them in fp32 storage (tsii_set_gemm_products(1)) --
  the arithmetic class config 5 asks for -- measured in the same test on the same weights.
"""
import json
import os

import numpy as np
import pytest
import torch

import text_segmentation_image_inpainting_amd as T
from oracle.filler import fill_state_dict_
from tests.backends import BACKENDS, both_backends
from text_segmentation_image_inpainting_amd import _lib, ops
from text_segmentation_image_inpainting_amd.BaseModels import to_nchw, to_nhwc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
BF16 = torch.bfloat16


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def run_block(m, x_nchw, gy_nchw, storage):
    """forward + backward of a block on an fp32 NCHW input, with the tensors inside the block in `storage`"""
    for p in m.parameters():
        p.grad = None
    x = x_nchw.clone().requires_grad_(True)
    xin = to_nchw(ops.to_storage(to_nhwc(x), storage))
    y = m(xin)
    y32 = to_nchw(ops.to_storage(to_nhwc(y), torch.float32))
    y32.backward(gy_nchw)
    return y32.detach(), x.grad.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


BLOCKS = [
    ("res_s2", lambda act: T.ResidualBlock(64, 128, 3, stride=2, padding=1, dilation=1, bias=False, BN=True, activation=act), (2, 64, 20, 20)),
    ("res_d2", lambda act: T.ResidualBlock(64, 64, 3, stride=1, padding=2, dilation=2, bias=False, BN=True, activation=act), (2, 64, 16, 16)),
    ("res_d4", lambda act: T.ResidualBlock(32, 32, 3, stride=1, padding=4, dilation=4, bias=False, BN=True, activation=act), (1, 32, 24, 24)),
    ("asp", lambda act: T.ASP(32, 32, act_fn=act, asp_rate=(3, 5, 9)), (2, 32, 14, 14)),
]


def rms(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).pow(2).mean().sqrt() / max(float(b.pow(2).mean().sqrt()), 1e-30))


@both_backends
@pytest.mark.parametrize("name,make,shape", BLOCKS, ids=[b[0] for b in BLOCKS])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("kinked", [False, True], ids=["smooth", "leaky"])
def test_blocks_in_bf16_storage(backend, name, make, shape, training, kinked):
    """smooth: the block with its activations removed -- every kernel and fusion of the bf16 path, nothing piecewise: EVERY tensor
    (output, dX, each weight gradient) within 1.5e-2 / 3e-2 of the fp32-storage result (measured: 4-7e-3 everywhere = sqrt(layers) x
    2^-9).  leaky: LeakyReLU(0.3) as the nets use it -- the outputs hold the same bar, the gradients cannot: a forward value
    perturbed by 2^-9 flips the side of the kink for ~0.4 % of the activations per layer and each flip changes that element's
    derivative by 70 %, i.e. ~4e-2 RMS per activation layer whatever the kernels do (measured 2-6e-2 RMS, 5-20e-2 max); the
    bar there is a sanity bound on that noise, the smooth variant is the one that pins the kernels."""
    with BACKENDS[backend]() as dev:
        torch.manual_seed(0)
        m = make(torch.nn.LeakyReLU(0.3) if kinked else None)
        fill_state_dict_(m.state_dict(), seed=7, gain=1.0)
        m = m.to(dev).train(training)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(shape, generator=g).to(dev)
        with torch.no_grad():
            yshape = m(x).shape
        gy = torch.randn(yshape, generator=g).to(dev)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        y32, dx32, g32 = run_block(m, x, gy, torch.float32)
        sd32 = {k: v.clone() for k, v in m.state_dict().items()}
        m.load_state_dict(sd0)
        y16, dx16, g16 = run_block(m, x, gy, BF16)
        assert set(g16) == set(g32)
        gmax = max(float(v.abs().max()) for v in g32.values())

        def relf(a, b):
            # a bias in front of a train-mode BatchNorm has an analytically zero gradient: in fp32 it is a sum of fp32 rounding
            # errors (< 1e-4 of the block's largest gradient), in bf16 storage a sum of bf16 ones -- held to a tenth of that scale
            # (other BatchNorm parameters in front of a per-channel conv + train-mode BatchNorm are NEARLY invariant -- only the zero
            # padding breaks it -- so their small gradients are differences of large sums: errors are measured against 5 % of gmax)
            if float(b.abs().max()) < 1e-4 * gmax:
                return 0.0 if float(a.abs().max()) <= 0.1 * gmax else float("inf")
            return float((a.double() - b.double()).abs().max() / max(float(b.abs().max()), 5e-2 * gmax))
        e_y, e_dx = rel(y16, y32), rel(dx16, dx32)
        worst = max((relf(g16[k], g32[k]), k) for k in g32)
        print(f"[bf16 storage] {name} {'train' if training else 'eval'} {'leaky' if kinked else 'smooth'}: y {e_y:.2e}  dx {e_dx:.2e} (rms {rms(dx16, dx32):.2e})  worst dW {worst[0]:.2e} ({worst[1]})")
        assert e_y <= 1.5e-2
        if kinked:
            assert rms(dx16, dx32) <= 0.15 and e_dx <= 0.5
            for k in g32:
                assert relf(g16[k], g32[k]) <= 0.5, (k, relf(g16[k], g32[k]))
        else:
            assert e_dx <= 1.5e-2
            # train mode: a BatchNorm's (gamma, beta) in front of conv + train-mode BatchNorm are nearly invariant directions -- their
            # gradients are small differences of large sums and carry the rounding noise of the terms (measured up to 3.9e-2)
            bar = 8e-2 if training else 3e-2
            for k in g32:
                assert relf(g16[k], g32[k]) <= bar, (k, relf(g16[k], g32[k]))
        if training:       # running statistics moved the same way
            sd16 = m.state_dict()
            for k in sd32:
                if "running" in k:
                    assert rel(sd16[k], sd32[k]) <= 1e-2, k


@both_backends
def test_stem_and_logits_head_bf16_storage(backend):
    """The two ends of a bf16 net: the fp32 image enters through the stem's space-to-depth rearrangement, the 1-channel logits
    leave as fp32 (head padded to 8 channels inside); and the partial-convolution family refuses bf16 tensors loudly."""
    from text_segmentation_image_inpainting_amd.BaseModels import Conv2d, ConvSpec, build_chain, run_chain

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            act = torch.nn.LeakyReLU(0.3)
            self.body = torch.nn.Sequential(*build_chain(3, (ConvSpec(32, 3, 2, 1), ConvSpec(16, 3, 1, 1)), act)[0], Conv2d(16, 1, 3, 1, 1))

        def forward(self, x):
            return run_chain(list(self.body), x)
    with BACKENDS[backend]() as dev:
        torch.manual_seed(0)
        m = Tiny()
        fill_state_dict_(m.state_dict(), seed=3, gain=1.0)
        m = m.to(dev).train()
        x = torch.randn(2, 3, 24, 20).to(dev)
        outs = {}
        try:
            for st in (torch.float32, BF16):
                T.set_activation_storage(st)
                for p in m.parameters():
                    p.grad = None
                y = m(x)
                assert y.dtype == torch.float32 and tuple(y.shape) == (2, 1, 12, 10)
                y.square().mean().backward()
                outs[st] = (y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
        finally:
            T.set_activation_storage(torch.float32)
        (y32, g32), (y16, g16) = outs[torch.float32], outs[BF16]
        assert rel(y16, y32) <= 2e-2
        gmax = max(float(v.abs().max()) for v in g32.values())
        for k in g32:       # two LeakyReLU layers: kink-flip noise (see test_blocks_in_bf16_storage)
            e = float((g16[k].double() - g32[k].double()).abs().max() / max(float(g32[k].abs().max()), 1e-3 * gmax))
            assert e <= 0.25, (k, e)
        # mask planes + bf16 tensors: no kernel, and no silent fp32 detour
        pc = T.PartialConv(8, 8, 3, 1, 1).to(dev)
        xb = ops.to_storage(torch.randn(1, 8, 8, 8).to(dev).permute(0, 2, 3, 1).contiguous(), BF16).permute(0, 3, 1, 2)
        with pytest.raises((NotImplementedError, RuntimeError)):
            pc((xb, torch.ones(1, 8, 8, 8).to(dev)))


def _seg_case(name, size):
    keys = json.load(open(os.path.join(GOLD, "seg_state_dict_keys.json")))[name]
    G = np.load(os.path.join(GOLD, f"{name.lower()}_{size}.npz"))
    return keys, G


def _net_run(m, x, t, storage, products=None, training=True):
    T.set_activation_storage(storage)
    if products is not None:
        _lib.set_gemm_products(products)
    try:
        for p in m.parameters():
            p.grad = None
        m.train(training)
        y = m(x)
        loss = T.BinaryFocalLoss(0, 1, 2)(y, t)
        loss.backward()
        torch.cuda.synchronize()
        return y.detach().clone(), float(loss.item()), {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    finally:
        T.set_activation_storage(torch.float32)
        if products is not None:
            _lib.set_gemm_products(None)


# net-level gradient rule of the reference-fixture test (train-mode BatchNorm, shipped LeakyReLU): a bf16-storage gradient is at most
# 2x as far from the reference's fp64 gradient as bf16 OPERANDS in fp32 storage put the same tensor (or 0.15), never beyond 0.75 of the
# tensor's maximum, and points the same way (cosine >= 0.85).  Measured on the chip (profiles/r06a_gputests_bf16.log): worst 0.551
# against 0.316 (ratio 1.74, the stem weight), lowest cosine 0.884.  Round 5 allowed 3x, an absolute 1.5 and no direction check.
COS_AB_RATIO = 2.0
ABS_CAP = 0.75
COS_MIN = 0.85


def _grad_errors(g16, g32):
    gmax = max(float(v.abs().max()) for v in g32.values())
    out = []
    for k in g32:
        den = max(float(g32[k].abs().max()), 1e-2 * gmax)
        out.append((float((g16[k].double() - g32[k].double()).abs().max()) / den, k))
    out.sort(reverse=True)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("training", [False, True], ids=["eval", "train"])
def test_xception_net_smooth_bf16_storage_every_gradient_256_gpu(training):
    """XceptionTextSegment 256^2, forward + backward, bf16 storage against fp32 storage with the net's LeakyReLU slope set to 1
    (same modules, same kernels and fusions -- load-time BatchNorm + LeakyReLU(slope) is the instruction sequence either way --
    but nothing piecewise): EVERY gradient tensor is bounded.  This is the test that pins the bf16 path's arithmetic at net level;
    with the real slope 0.3 the same comparison measures activation-kink flips, not kernels (next test)."""
    G = np.load(os.path.join(GOLD, "xceptiontextsegment_256.npz"))
    with BACKENDS["gpu"]() as dev:
        m = T.XceptionTextSegment()
        fill_state_dict_(m.state_dict(), seed=43, gain=1.0)
        for mod in m.modules():
            if isinstance(mod, torch.nn.LeakyReLU):
                mod.negative_slope = 1.0
        m = m.to(dev)
        x = torch.from_numpy(np.random.default_rng(int(G["seed_x"])).standard_normal((2, 3, 256, 256)).astype(np.float32)).to(dev)
        t = (torch.from_numpy(np.random.default_rng(int(G["seed_t"])).uniform(size=(2, 1, 256, 256))) > 0.8).float().to(dev)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        y32, l32, g32 = _net_run(m, x, t, torch.float32, training=training)
        m.load_state_dict(sd0)
        y16, l16, g16 = _net_run(m, x, t, BF16, training=training)
        e = rel(y16, y32)
        errs = _grad_errors(g16, g32)
        print(f"[bf16 storage] smooth net 256^2 {'train' if training else 'eval'}: out {e:.2e} loss {abs(l16 - l32):.2e}; "
              f"gradients worst {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}, tensors {len(errs)}")
        # measured on the chip: train 1.5e-2 / worst gradient 4.9e-2 / median 9.6e-3; eval (the filler's running statistics, no
        # renormalisation: values grow through the net) 2.6e-2 / 9.1e-2 / 1.8e-2
        assert e <= 4e-2 and abs(l16 - l32) <= 2e-2 * max(1.0, abs(l32))      # (eval with slope 1: logits of order 1e3, the loss is linear in them)
        for v, k in errs:
            assert v <= 0.15, (k, v)
        assert errs[len(errs) // 2][0] <= 3e-2


@pytest.mark.gpu
def test_xception_net_bf16_storage_vs_reference_fixture_256_gpu():
    """XceptionTextSegment 256^2 (cfg 1 size) as shipped (LeakyReLU 0.3) in bf16 storage against the fixture the REFERENCE produced
    in fp64 (tests/golden/make_golden_misc.py).  Outputs and loss: absolute bars.  Gradients carry activation-kink flips (a forward
    value moved by 2^-9 changes the side of the kink for ~0.4 % of the activations per layer, each flip a 70 % change of that
    element's derivative) on top of the train-mode BatchNorm amplification every rounding source suffers in this net
    (SURVEY.md F11): every recorded tensor is bounded (a) absolutely and (b) relative to what bf16-rounded OPERANDS in fp32 storage
    (tsii_set_gemm_products(1), the arithmetic class of config 5 without the storage) do to the same tensor in the same test."""
    name = "XceptionTextSegment"
    keys, G = _seg_case(name, 256)
    with BACKENDS["gpu"]() as dev:
        m = getattr(T, name)()
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == keys
        fill_state_dict_(m.state_dict(), seed=43, gain=1.0)
        m = m.to(dev)
        x = torch.from_numpy(np.random.default_rng(int(G["seed_x"])).standard_normal((2, 3, 256, 256)).astype(np.float32)).to(dev)
        t = (torch.from_numpy(np.random.default_rng(int(G["seed_t"])).uniform(size=(2, 1, 256, 256))) > 0.8).float().to(dev)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        # --- eval-mode BatchNorm against the reference's fp64 eval output, gradients against fp32 storage
        y32, l32, g32 = _net_run(m, x, t, torch.float32, training=False)
        y16, l16, g16 = _net_run(m, x, t, BF16, training=False)
        e = rel(y16, torch.from_numpy(G["y_eval_f64"]))
        errs = _grad_errors(g16, g32)
        print(f"[bf16 storage] eval-mode BN 256^2: out vs the reference's fp64 run {e:.2e}, loss vs fp32 storage {abs(l16 - l32):.2e}; "
              f"gradients vs fp32 storage worst {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}")
        assert e <= 3e-2 and abs(l16 - l32) <= 5e-3
        for v, k in errs:
            assert v <= 0.35, (k, v)
        assert errs[len(errs) // 2][0] <= 4e-2
        # --- train-mode BatchNorm against the reference's fp64 run
        y64 = torch.from_numpy(G["y_train_f64"])
        m.load_state_dict(sd0)
        y16, l16, g16 = _net_run(m, x, t, BF16)
        m.load_state_dict(sd0)
        yp1, lp1, gp1 = _net_run(m, x, t, torch.float32, products=1)
        e16, ep1 = rel(y16, y64), rel(yp1, y64)
        print(f"[bf16 storage] train-mode BN 256^2 vs the reference's fp64 run: out {e16:.2e} (bf16 operands in fp32 storage: {ep1:.2e}); loss {abs(l16 - float(G['loss_f64'])):.2e}")
        assert e16 <= 8e-2 and e16 <= max(5e-2, 2.5 * ep1) and abs(l16 - float(G["loss_f64"])) <= 5e-3      # measured 4.96e-2 | 3.53e-2
        rows = []
        for k in G.files:
            if not k.startswith("grad64."):
                continue
            kk = k[7:]
            r64 = torch.from_numpy(G[k])
            cos = float(torch.nn.functional.cosine_similarity(g16[kk].double().flatten().cpu(), r64.double().flatten(), dim=0))
            rows.append((rel(g16[kk], r64), rel(gp1[kk], r64), kk, cos))
        assert len(rows) >= 20
        rows.sort(reverse=True)
        print("[bf16 storage] train-mode gradients vs fp64 (bf16 storage | bf16 operands, fp32 storage | cosine with fp64):")
        for a_, b_, kk, c_ in rows[:6]:
            print(f"    {kk}: {a_:.3f} | {b_:.3f} | {c_:.3f}")
        med16, medp1 = rows[len(rows) // 2][0], sorted(r[1] for r in rows)[len(rows) // 2]
        print(f"    median {med16:.3f} | {medp1:.3f}; worst ratio a/b {max(r[0] / max(r[1], 1e-9) for r in rows):.2f}, lowest cosine {min(r[3] for r in rows):.3f}")
        for a_, b_, kk, c_ in rows:
            # every tensor: bounded by what operand rounding alone does to it in the same test, capped, and pointing the same way
            assert a_ <= max(0.15, COS_AB_RATIO * b_) and a_ <= ABS_CAP, (kk, a_, b_)
            assert c_ >= COS_MIN, (kk, c_)
        assert med16 <= max(0.1, 2.5 * medp1)


@pytest.mark.gpu
def test_xception_1024_bs8_bf16_storage_properties_gpu():
    """cfg 5 at ITS size (1024^2, 8 images, bf16 storage): determinism (bit-identical repeats), batch independence in eval mode,
    finite and bit-repeatable gradients, and agreement with the fp32-storage run of the same weights."""
    with BACKENDS["gpu"]() as dev:
        torch.manual_seed(0)
        m = T.XceptionTextSegment().to(dev)
        from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
        x, t = (v.to(dev) for v in make_seg_batch(8, 1024, seed0=0))
        try:
            T.set_activation_storage(BF16)
            m.eval()
            with torch.no_grad():
                ye = m(x)
                ye2 = m(x)
                y1 = m(x[5:6])
            assert torch.equal(ye, ye2)
            assert rel(ye[5:6], y1) <= 1e-6          # an image's logits do not depend on its batch mates (eval-mode BatchNorm)
            T.set_activation_storage(torch.float32)
            with torch.no_grad():
                ye32 = m(x)
            e = rel(ye, ye32)
            print(f"[bf16 storage] 1024^2 bs 8 eval output vs fp32 storage: {e:.2e}")
            assert e <= 5e-2
            del ye, ye2, y1, ye32
            grads = []
            for _ in range(2):
                T.set_activation_storage(BF16)
                sd = {k: v.clone() for k, v in m.state_dict().items()}
                for p in m.parameters():
                    p.grad = None
                m.train()
                loss = T.BinaryFocalLoss(0, 1, 2)(m(x), t)
                loss.backward()
                torch.cuda.synchronize()
                grads.append((float(loss.item()), [p.grad.clone() for p in m.parameters()]))
                m.load_state_dict(sd)
            assert grads[0][0] == grads[1][0] and np.isfinite(grads[0][0])
            for a, b in zip(grads[0][1], grads[1][1]):
                assert torch.isfinite(a).all() and torch.equal(a, b)
            print(f"[bf16 storage] 1024^2 bs 8 train step: loss {grads[0][0]:.5f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        finally:
            T.set_activation_storage(torch.float32)


# band of the training-curve test below, set from the chip's own print-out (profiles/r06h_gputests_bf16_curve.log)
CURVE_REL_BAND = 0.02        # |loss_bf16 - loss_fp32| <= 2 % of the fp32-storage loss at every step (measured worst 0.64 %) ...
CURVE_END_BAND = 0.01        # ... and 1 % averaged over the last 10 steps (measured 0.11 %)


@pytest.mark.gpu
def test_xception_bf16_storage_trains_like_fp32_storage_50_steps_gpu():
    """Does cfg 5's arithmetic TRAIN?  XceptionTextSegment at 256^2, 4 images per step, 50 SGD-Nesterov steps on a fixed seeded stream
    of synthetic batches (fresh batch every step), the same initial weights once in fp32 storage and once in bf16 storage: the two loss
    curves must stay within a stated band of each other at every step and both must go down.  Per-tensor gradient distances of a
    single step (0.13 median of the tensor maximum in train mode, previous tests) are activation-kink noise; what matters for a user
    is that the optimisation follows the same path at the scale of the loss."""
    from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    with BACKENDS["gpu"]() as dev:
        batches = [tuple(v.to(dev) for v in make_seg_batch(4, 256, seed0=1000 + 4 * i)) for i in range(50)]
        curves = {}
        try:
            for storage in ("f32", "bf16"):
                T.set_activation_storage(storage)
                torch.manual_seed(0)
                net = T.XceptionTextSegment().to(dev).train()

                class Step(torch.nn.Module):
                    def __init__(self, net):
                        super().__init__()
                        self.net = net

                    def forward(self, a):
                        return self.net(a[0])
                focal = T.BinaryFocalLoss(0, 1, 2)
                tr = FlatSGDTrainer(Step(net), lr=2e-3, momentum=0.9, weight_decay=1e-4, loss_fn=lambda out, tgt: focal(out, tgt))
                curves[storage] = [float(tr.step(x, None, t)) for x, t in batches]
                tr.close()
        finally:
            T.set_activation_storage("f32")
        a, b = np.array(curves["f32"]), np.array(curves["bf16"])
        dev_rel = np.abs(b - a) / np.maximum(np.abs(a), 1e-6)
        print("[bf16 storage] 50-step training curves (fp32 storage | bf16 storage), every 5th step:")
        for i in range(0, 50, 5):
            print(f"    step {i:2d}: {a[i]:.5f} | {b[i]:.5f}")
        print(f"    last: {a[-1]:.5f} | {b[-1]:.5f}; worst relative distance {dev_rel.max():.3e} (step {int(dev_rel.argmax())}), "
              f"mean over the last 10 steps {abs(b[-10:].mean() - a[-10:].mean()) / a[-10:].mean():.3e}")
        assert np.isfinite(a).all() and np.isfinite(b).all()
        assert a[-10:].mean() < 0.9 * a[:5].mean() and b[-10:].mean() < 0.9 * b[:5].mean(), "both runs must make progress"
        assert dev_rel.max() <= CURVE_REL_BAND, (float(dev_rel.max()), int(dev_rel.argmax()))
        assert abs(b[-10:].mean() - a[-10:].mean()) <= CURVE_END_BAND * a[-10:].mean()

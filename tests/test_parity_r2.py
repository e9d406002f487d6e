"""Round-2 parity holes (VERDICT r1 "What's weak" 1-5):

* large-dilation depth-wise convolutions (d = 8 / 16 / 17 / 29: MobileNetV2 stage 6/7, RFB branches,
  models/MobileNetV2.py:203-215, models/common.py:102) on maps where every off-centre tap is IN range;
* RFB on a 32x32 map and TextSegament / XceptionTextSegment at 256x256 (32x32 map at 1/8) vs the CPU oracle;
* the fused SGD-Nesterov kernel vs ``torch.optim.SGD(momentum, nesterov=True, weight_decay)`` (checkpoints/ReadME.md:4);
* ``DoubleUpSample`` / up-sample + concat as exact copies (models/partial_convolution.py:229-231);
* BASELINE config 3 at full size (TextSegament 512x512): determinism, batch independence, finite gradients.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import text_segmentation_image_inpainting_amd as T
from oracle import pconv_oracle as O
from oracle import seg_oracle as S
from oracle.filler import fill_state_dict_, make_state_dict
from tests.backends import BACKENDS, both_backends
from tests.util import low_rank_error, assert_close, rel_err

TOL = 1e-3
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@both_backends
def test_depthwise_large_dilation_taps_in_range(backend):
    """d = 8/16/17/29 with H, W > 2d: all nine taps land inside the map for the interior pixels (the 64x64-input
    fixtures only ever exercised the centre tap).  Plain depth-wise conv (BaseModels.Conv2d, the segmentation path) and
    the depth-wise PartialConv with a hole mask; forward, dX, dW (db) vs stock torch / the oracle on CPU."""
    from text_segmentation_image_inpainting_amd.BaseModels import Conv2d
    cases = [(8, 8, 33, 40), (8, 16, 64, 64), (12, 17, 40, 64), (8, 29, 64, 61), (8, 29, 64, 64)]
    with BACKENDS[backend]() as dev:
        for idx, (c, d, H, W) in enumerate(cases):
            rng = np.random.default_rng(1200 + idx)
            conv = Conv2d(c, c, 3, 1, d, d, groups=c, bias=True)
            fill_state_dict_(conv.state_dict(), seed=1200 + idx)
            x = torch.from_numpy(rng.standard_normal((2, c, H, W)).astype(np.float32))
            w = conv.weight.detach().clone().requires_grad_(True)
            b = conv.bias.detach().clone().requires_grad_(True)
            xo = x.clone().requires_grad_(True)
            yo = F.conv2d(xo, w, b, 1, d, d, c)
            gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            conv = conv.to(dev)
            xd = x.to(dev).requires_grad_(True)
            y = conv(xd)
            assert_close(y, yo, TOL, f"dw d={d} y")
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, f"dw d={d} dx")
            assert_close(conv.weight.grad, w.grad, TOL, f"dw d={d} dw")
            assert_close(conv.bias.grad, b.grad, TOL, f"dw d={d} db")
        for idx, (c, d, H, W) in enumerate([(8, 16, 48, 56), (8, 8, 40, 33)]):
            rng = np.random.default_rng(1300 + idx)
            m = T.PartialConv(c, c, 3, 1, d, d, c, True, True)
            fill_state_dict_(m.state_dict(), seed=1300 + idx)
            x = torch.from_numpy(rng.standard_normal((2, c, H, W)).astype(np.float32))
            pa = (torch.from_numpy(rng.uniform(size=(2, 1, H, W))) > 0.3).float()
            pa[:, :, 5:30, 7:29] = 0        # a hole wider than the dilation: some windows are all-hole
            mask = pa.expand(-1, c, -1, -1).contiguous()
            w = m.feature_conv.weight.detach().clone().requires_grad_(True)
            b = m.feature_conv.bias.detach().clone().requires_grad_(True)
            xo = x.clone().requires_grad_(True)
            yo, nmo = O.partial_conv(xo, mask, w, b, 1, d, d, c, True)
            gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            m = m.to(dev)
            xd = x.to(dev).requires_grad_(True)
            y, nm = m((xd, pa.expand(-1, c, -1, -1).to(dev)))
            assert_close(y, yo, TOL, f"pconv dw d={d} y")
            assert np.array_equal(nm.detach().cpu().numpy(), nmo.detach().numpy()), f"pconv dw d={d} new_mask"
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, f"pconv dw d={d} dx")
            assert_close(m.feature_conv.weight.grad, w.grad, TOL, f"pconv dw d={d} dw")


def _rfb_case(dev, hw, cin, cout, seed, report=False, smooth=False, coarse_bar=None):
    """RFB forward / dX / every parameter gradient vs the oracle.  Train-mode BatchNorm over few samples makes some of
    these gradients ill-conditioned (at TextSegament's channel counts the oracle's own fp32 run is 1.3e-2 away from its
    fp64 run in dX), so the target is the fp64 oracle and the tolerance per tensor is max(floor, 4x the oracle's
    fp32-vs-fp64 discrepancy).

    Two forms.  ``smooth`` (no activation: every kernel of the block, nothing piecewise): EVERY tensor within the bar --
    this is the check of the kernels.  LeakyReLU (the form the nets use): two fp32 implementations put about one
    activation in a million on different sides of the kink and one flip moves the gradients of the layers around it by
    several 1e-3 (tests/util.py: assert_gradients_close; seen here as 6.4e-3 on one (k,1) weight after the 1x1 GEMMs moved
    to another -- equally exact, 1e-6 vs fp64 per call -- kernel), so that form allows a few tensors up to 5x the bar and
    holds the median to a tenth of it."""
    act = None if smooth else torch.nn.LeakyReLU(0.3)
    m = T.RFB(cin, cout, activation=act, add_sece=True)
    fill_state_dict_(m.state_dict(), seed=seed)
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((2, cin, hw, hw)).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((2, cout, hw, hw)).astype(np.float32))

    def oracle(dtype):
        sd = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in m.state_dict().items()}
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        xo = x.to(dtype).clone().requires_grad_(True)
        yo = S.rfb(sd, "", xo, cout, (lambda t: t) if smooth else O.leaky(0.3), True)
        yo.backward(gy.to(dtype))
        return yo.detach(), xo.grad, {k: v.grad for k, v in sd.items() if v.grad is not None}

    y32, dx32, g32 = oracle(torch.float32)
    y64, dx64, g64 = oracle(torch.float64)
    m = m.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    y = m(xd)
    if coarse_bar is not None:
        # a coarser arithmetic (gemm products = 1: bf16-rounded operands) on the kink-free form: train-mode forward, dX and
        # every parameter gradient within ``coarse_bar`` of the fp64 oracle -> (worst forward, worst backward)
        assert smooth
        e_y = assert_close(y, y64, coarse_bar, f"RFB {hw}x{hw} y (coarse arithmetic)")
        y.backward(gy.to(dev))
        e_b = assert_close(xd.grad, dx64, coarse_bar, f"RFB {hw}x{hw} dx (coarse arithmetic)")
        gmax = max(float(v.abs().max()) for v in g64.values())
        for k, p in m.named_parameters():
            e_b = max(e_b, assert_close(p.grad, g64[k], coarse_bar, f"RFB {hw}x{hw} grad {k} (coarse arithmetic)", floor=1e-3 * gmax))
        return e_y, e_b
    assert_close(y, y64, max(TOL, 4 * rel_err(y32, y64)), f"RFB {hw}x{hw} y")
    y.backward(gy.to(dev))
    assert_close(xd.grad, dx64, max(2e-3, 4 * rel_err(dx32, dx64)), f"RFB {hw}x{hw} dx")
    params = dict(m.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values())
    rows = []
    for k, ref in g64.items():
        floor = 1e-3 * gmax
        noise = rel_err(g32[k], ref, floor)
        e = rel_err(params[k].grad, ref, floor)
        rows.append((e / max(noise, 7.5e-4), k, e, noise))
    rows.sort(reverse=True)
    if report:
        print(f"\n[RFB {cin}->{cout} {hw}x{hw}] dx err {rel_err(xd.grad, dx64):.2e} (oracle fp32-vs-fp64 {rel_err(dx32, dx64):.2e}); worst gradient tensors:")
        for r in rows[:6]:
            print(f"   ratio {r[0]:6.2f}  {r[1]:40s} err {r[2]:.2e}  oracle fp32 noise {r[3]:.2e}")
    over = [(e, k) for ratio, k, e, noise in rows if e > max(3e-3, 4 * noise)]
    if smooth:
        assert not over, f"RFB {hw}x{hw} (no activation): {over[:5]}"
    else:
        for ratio, k, e, noise in rows:
            assert e <= 5 * max(3e-3, 4 * noise), f"RFB {hw}x{hw} grad {k}: {e:.2e} vs fp32 noise {noise:.2e}"
        assert len(over) <= max(4, len(rows) // 25), f"RFB {hw}x{hw}: {len(over)} tensors beyond the bar: {over[:8]}"
        errs = sorted(e for _, _, e, _ in rows)
        assert errs[len(errs) // 2] <= 3e-4, f"RFB {hw}x{hw}: median gradient error {errs[len(errs) // 2]:.2e}"
    assert len(rows) >= 20


def test_rfb_32x32_emu():
    """RFB with the d = 5 / 17 taps in range (32x32 map = cfg 1's 1/8 map), small channel counts, through the emulator."""
    with BACKENDS["emu"]() as dev:
        _rfb_case(dev, 32, 16, 8, seed=1400)
        _rfb_case(dev, 32, 16, 8, seed=1400, smooth=True)


@pytest.mark.gpu
def test_rfb_64x64_gpu(capsys):
    """RFB at cfg 3's 64x64 map: every branch dilation (5 / 17 / 29) has live off-centre taps."""
    with BACKENDS["gpu"]() as dev, capsys.disabled():
        _rfb_case(dev, 64, 64, 32, seed=1401)
        _rfb_case(dev, 32, 48, 16, seed=1402)
        _rfb_case(dev, 32, 1344, 256, seed=1403, report=True)      # TextSegament's own RFB (in 1344, out 256) on cfg 1's 32x32 map
        _rfb_case(dev, 32, 1344, 256, seed=1403, report=True, smooth=True)
        _rfb_case(dev, 64, 64, 32, seed=1401, smooth=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["TextSegament", "XceptionTextSegment"])
def test_seg_nets_256_vs_oracle_gpu(name, capsys):
    """cfg 1 size (256x256 -> 32x32 map at 1/8): eval and train forward, focal loss and every trainable gradient vs
    the CPU oracle (itself pinned to the reference at 64x64 by tests/test_oracle_seg_golden.py).

    Train-mode gradients of these ~100-layer BatchNorm nets are chaotic at the fp32 rounding level: a
    1-ulp perturbation of the input moves some of the ORACLE's own fp32 gradient tensors by 1-2e-2 of their maximum
    (LeakyReLU kinks under train-mode BatchNorm; the RFB alone, at these channel counts, has an fp32-vs-fp64 discrepancy
    of 1.3e-2 in dX on the CPU -- the HIP RFB is at 4e-4 of the fp64 result, see test_rfb_64x64_gpu).  A kernel bug
    gives O(1) errors on the affected tensors; rounding chaos gives a heavy-tailed few-percent scatter.  The yardstick is
    therefore the oracle's own fp32 noise per tensor (largest deviation from the fp64 gradient over its plain fp32 run and
    three 1-ulp input perturbation runs), and the bars are: every tensor within 16x (max error) / 4x (RMS) of that noise (floors 3e-3 / 1e-3; observed worst 8.2x / 2.3x), and the
    MEDIAN tensor within 2x -- i.e. the bulk agrees at noise level and nothing is off by more than the heavy tail
    allows.  The worst tensors are printed with their ratios."""
    keys = json.load(open(os.path.join(GOLD, "seg_state_dict_keys.json")))[name]
    fn = S.SEG_MODELS[name]
    rng = np.random.default_rng(1500)
    x = torch.from_numpy(rng.standard_normal((2, 3, 256, 256)).astype(np.float32))
    t = (torch.from_numpy(rng.uniform(size=(2, 1, 256, 256))) > 0.8).float()

    def oracle(dtype):
        sd = make_state_dict([(k, s) for k, s in keys], seed=43, gain=1.0, dtype=dtype)
        with torch.no_grad():
            ye = fn(sd, x.to(dtype), training=False)
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        y = fn(sd, x.to(dtype), training=True)
        loss = S.binary_focal_loss(y, t.to(dtype), 0.0, 1.0, 2.0)
        loss.backward()
        return ye, y.detach(), float(loss), {k: v.grad for k, v in sd.items() if v.grad is not None}

    ye32, y32, l32, g32 = oracle(torch.float32)
    ye64, y64, l64, g64 = oracle(torch.float64)
    # further samples of the fp32 noise: the oracle's fp32 gradients under 1-ulp (1e-7 relative) perturbations of the input.
    # Measured on the CPU: the error of rfb.3.4.weight against the fp64 gradient across five such runs is
    # 7.7e-3, 7.8e-3, 2.9e-2, 1.3e-4, 7.6e-3 -- discrete jumps (an activation crossing a LeakyReLU kink), not a smooth spread.
    x_keep, gperts = x, []
    for sample in range(3):
        x = x_keep * (1 + 1e-7 * torch.from_numpy(np.random.default_rng(100 + sample).standard_normal(tuple(x_keep.shape)).astype(np.float32)))
        gperts.append(oracle(torch.float32)[3])
    x = x_keep
    with BACKENDS["gpu"]() as dev:
        m = getattr(T, name)()
        fill_state_dict_(m.state_dict(), seed=43, gain=1.0)
        m = m.to(dev)
        m.eval()
        with torch.no_grad():
            ye = m(x.to(dev))
        assert_close(ye, ye64, TOL, name + " 256 eval vs fp64 oracle")
        m.train()
        y = m(x.to(dev))
        noise_y = float((y32.double() - y64).abs().max() / y64.abs().max())
        assert_close(y, y64, max(TOL, 4 * noise_y), name + " 256 train vs fp64 oracle")
        loss = T.BinaryFocalLoss(0, 1, 2)(y, t.to(dev))
        assert abs(loss.item() - l64) < 1e-4 * max(1.0, abs(l64))
        loss.backward()
        params = dict(m.named_parameters())
        gmax = max(float(v.abs().max()) for v in g64.values())
        rows, bad, stat = [], [], {}
        for k, ref64 in g64.items():
            ours = params[k].grad.detach().cpu().double()
            scale = max(float(ref64.abs().max()), 1e-3 * gmax)
            rscale = max(float(ref64.pow(2).mean().sqrt()), 1e-3 * gmax)
            devs = [g32[k].double() - ref64] + [gp[k].double() - ref64 for gp in gperts]
            n_max = max(float(d.abs().max()) for d in devs) / scale
            n_rms = max(float(d.pow(2).mean().sqrt()) for d in devs) / rscale
            e_max = float((ours - ref64).abs().max()) / scale
            e_rms = float((ours - ref64).pow(2).mean().sqrt()) / rscale
            stat[k] = (e_max, n_max, e_rms, n_rms)
        # Four fp32 runs are four samples of a heavy-tailed noise: which tensors a run's kink flips hit differs from run to run
        # (round 4: a change that left every kernel output of this net bit-identical and moved only the summation order of the
        # BatchNorm statistics -- 1e-7 -- moved THIS run's 2e-2 .. 4e-2 outliers from rfb.3.* onto rfb_linear_conv, whose own
        # four samples happened to be quiet; profiles/r04k_seg256_ab.log, r04l_dw_ab_textsegament.log).  A flip inside a module
        # perturbs every gradient of that module and of everything upstream, so a tensor's noise is taken no smaller than the
        # worst the oracle showed anywhere in its top-level module.
        fam_max, fam_rms = {}, {}
        for k, (e_max, n_max, e_rms, n_rms) in stat.items():
            f = k.split(".")[0]
            fam_max[f] = max(fam_max.get(f, 0.0), n_max)
            fam_rms[f] = max(fam_rms.get(f, 0.0), n_rms)
        for k, (e_max, n_max, e_rms, n_rms) in stat.items():
            f = k.split(".")[0]
            rows.append((e_max / max(n_max, 3e-4), e_rms / max(n_rms, 2.5e-4), k, e_max, n_max, e_rms, n_rms))
            if e_max > max(3e-3, 16 * max(n_max, fam_max[f])) or e_rms > max(1e-3, 4 * max(n_rms, fam_rms[f])):
                bad.append((k, e_max, n_max, e_rms, n_rms, fam_max[f], fam_rms[f]))
        rows.sort(reverse=True)
        median_ratio = float(np.median([r[0] for r in rows]))
        with capsys.disabled():
            print(f"\n[{name} 256 grads] {len(g64)} tensors; error vs the fp64 gradient relative to the oracle's own fp32 noise "
                  f"(worst of 4 fp32 oracle runs); worst tensors:")
            for r in rows[:8]:
                print(f"   max-ratio {r[0]:7.2f}  rms-ratio {r[1]:6.2f}  {r[2]:60s} e_max {r[3]:.2e} n_max {r[4]:.2e} e_rms {r[5]:.2e} n_rms {r[6]:.2e}")
            print(f"   median max-ratio over all tensors: {median_ratio:.2f}; worst oracle noise per module (max / rms): " +
                  ", ".join(f"{f} {fam_max[f]:.1e} / {fam_rms[f]:.1e}" for f in sorted(fam_max)))
        assert not bad, bad[:5]
        assert median_ratio <= 2.0
        assert len(g64) >= 100


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["TextSegament", "XceptionTextSegment"])
def test_seg_nets_256_vs_reference_fixture_gpu(name, capsys):
    """cfg 1 size against what the REFERENCE ITSELF produced (tests/golden/<name>_256.npz, written by
    tests/golden/make_golden_misc.py from /root/reference/models/text_segmentation.py + loss.py in fp32 and fp64; the inputs are
    rebuilt from the recorded seeds): eval output and train output at 1e-3 / noise level, focal loss at 1e-4, and the recorded
    sample of 24 gradient tensors.  The gradient yardstick is the reference's own fp32-vs-fp64 discrepancy, the largest over THREE
    fp32 runs of the reference (round 5: the plain one and two with the input moved by one ulp -- a heavy-tailed noise, see
    test_seg_nets_256_vs_oracle_gpu), and no smaller than the median over the recorded tensors.  Bars: 16x that noise per tensor
    (floor 3e-3), median tensor within 3x, every tensor beyond 4x explained by the flip signature of tests/util.py (97 %; vectors:
    a confirmed flip -- a weight tensor with the signature, or >= 80 % of their own error in <= 3 entries -- and 2x the bar)."""
    G = np.load(os.path.join(GOLD, name.lower() + "_256.npz"))
    x = torch.from_numpy(np.random.default_rng(int(G["seed_x"])).standard_normal((2, 3, 256, 256)).astype(np.float32))
    t = (torch.from_numpy(np.random.default_rng(int(G["seed_t"])).uniform(size=(2, 1, 256, 256))) > 0.8).float()
    with BACKENDS["gpu"]() as dev:
        m = getattr(T, name)()
        fill_state_dict_(m.state_dict(), seed=43, gain=1.0)
        m = m.to(dev)
        m.eval()
        with torch.no_grad():
            ye = m(x.to(dev))
        assert_close(ye, G["y_eval_f64"], TOL, name + " 256 eval vs the reference's fp64 run")
        assert_close(ye, G["y_eval"], TOL, name + " 256 eval vs the reference's fp32 run")
        m.train()
        y = m(x.to(dev))
        noise_y = float(np.abs(G["y_train"] - G["y_train_f64"]).max() / np.abs(G["y_train_f64"]).max())
        assert_close(y, G["y_train_f64"], max(TOL, 4 * noise_y), name + " 256 train vs the reference's fp64 run")
        loss = T.BinaryFocalLoss(0, 1, 2)(y, t.to(dev))
        assert abs(loss.item() - float(G["loss_f64"])) < 1e-4
        loss.backward()
        params = dict(m.named_parameters())
        names = [k[5:] for k in G.files if k.startswith("grad.")]
        assert len(names) >= 20
        gmax = max(float(np.abs(G["grad64." + k]).max()) for k in names)
        noise, err = {}, {}
        for k in names:
            ref64 = G["grad64." + k].astype(np.float64)
            scale = max(float(np.abs(ref64).max()), 1e-3 * gmax)
            # the reference's own fp32 noise for this tensor: the largest of THREE fp32 runs of the reference (plain and two inputs moved
            # by one ulp; tests/golden/make_golden_misc.py records their distances from its fp64 run) -- a heavy-tailed quantity
            noise[k] = float(np.max(G["noise3." + k])) * float(np.abs(ref64).max()) / scale
            err[k] = float(np.abs(params[k].grad.detach().cpu().double().numpy() - ref64).max()) / scale
        pooled = float(np.median(list(noise.values())))
        rows = sorted(((err[k] / max(noise[k], pooled, 3e-4), k, err[k], noise[k]) for k in names), reverse=True)
        with capsys.disabled():
            print(f"\n[{name} 256 vs reference fixture] {len(names)} gradient tensors, pooled fp32 noise of the reference {pooled:.2e}; worst:")
            for r in rows[:6]:
                print(f"   ratio {r[0]:6.2f}  {r[1]:60s} err {r[2]:.2e}  reference fp32-vs-fp64 {r[3]:.2e}")
        for ratio, k, e, n in rows:
            assert e <= max(3e-3, 16 * max(n, pooled)), (k, e, n, pooled)
        assert float(np.median([r[0] for r in rows])) <= 3.0
        # Outliers (beyond 4x the noise) must be EXPLAINED by the rule of tests/util.py: a weight gradient carries >= 97 % of its squared
        # error in <= 3 singular values (single activation-kink flips; a wrong kernel gives a dense error).  A bias / BatchNorm vector
        # has no singular values to show.  Round 6: it may ride -- up to 2x the outlier bar -- only on evidence from ITS OWN LAYER'S
        # WEIGHT gradient, which the fixture now records for every sampled vector ("partner.<vector>" -> the convolution the bias
        # belongs to / the BatchNorm normalises, tests/golden/make_golden_misc.py: partners_256).  A flip behind that BatchNorm at
        # channel c moves entry c of its bias / weight gradients AND row c of the convolution's weight gradient (a different kernel: the
        # dW product, not the BatchNorm reduction), so: (a) the vector's error sits in <= 3 entries (>= 80 % of it), and (b) the SAME
        # channels stand out in the partner's error -- each of them among the partner's 5 highest-energy rows and >= 3x the median row.
        # A dense BatchNorm-backward error has neither property; an error confined to the vector's kernel has (a) but not (b).  The
        # round-5 rule accepted (a) alone; that is gone.  Vectors whose partner is too large to record (> 400 000 elements) need a
        # confirmed weight tensor elsewhere in the same run, as before round 5.
        outliers = [(k, e) for ratio, k, e, n in rows if e > max(3e-3, 4 * max(n, pooled))]
        judged = {}
        for k, e in outliers:
            ok, f = low_rank_error(params[k].grad, G["grad64." + k].astype(np.float32), frac=0.97)
            vec = G["grad64." + k].squeeze().ndim <= 1
            partner = None
            if vec and ("partner." + k) in G.files and G["grad64." + k].size >= 64:
                w = str(G["partner." + k])
                ref = (G["grad64." + w] if ("grad64." + w) in G.files else G["partner64." + w]).astype(np.float64)
                ew = params[w].grad.detach().cpu().double().numpy() - ref
                rown = np.sqrt((ew.reshape(ew.shape[0], -1) ** 2).sum(1))
                ev = (params[k].grad.detach().cpu().double().numpy() - G["grad64." + k].astype(np.float64)).reshape(-1)
                top = np.argsort(-np.abs(ev))[:3]
                share = float((ev[top] ** 2).sum() / max((ev ** 2).sum(), 1e-300))
                # the entries that carry the vector's error (those above a tenth of the largest one, at most 3)
                chans = [int(c) for c in top if abs(ev[c]) >= 0.1 * abs(ev[top[0]])]
                rank = {c: int((rown > rown[c]).sum()) for c in chans}
                lift = {c: float(rown[c] / max(np.median(rown), 1e-300)) for c in chans}
                pok = share >= 0.8 and len(rown) == ev.size and all(rank[c] < 5 and lift[c] >= 3.0 for c in chans)
                partner = (pok, share, w, chans, rank, lift)
            judged[k] = (ok, f, vec, partner)
            with capsys.disabled():
                print(f"   outlier {k}: err {e:.2e}, {100 * f:.1f} % of it in <= 3 singular values / entries" +
                      (f"; channels {partner[3]} of its layer's weight {partner[2]}: row-energy ranks {partner[4]}, x median row {partner[5]}" if partner else ""))
        confirmed = any(ok and not vec for ok, f, vec, partner in judged.values())
        for k, (ok, f, vec, partner) in judged.items():
            rides = vec and (partner[0] if partner is not None else confirmed) and err[k] <= 2 * max(3e-3, 4 * max(noise[k], pooled))
            assert ok or rides, (k, f, err[k], partner, "dense gradient error beyond 4x the reference's own fp32 noise")


@pytest.mark.gpu
def test_sgd_nesterov_vs_torch_gpu():
    """tsii_sgd_nesterov over a flat buffer == torch.optim.SGD(lr, momentum, nesterov=True, weight_decay) on CPU,
    three steps with fresh gradients (first step: buf = grad; torch/optim/sgd.py semantics)."""
    from text_segmentation_image_inpainting_amd import ops
    rng = np.random.default_rng(1600)
    for lr, mom, wd, n in ((0.01, 0.9, 1e-4, 100003), (0.1, 0.8, 0.0, 4096), (0.05, 0.95, 1e-2, 777)):
        p0 = rng.standard_normal(n).astype(np.float32)
        grads = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
        pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = torch.optim.SGD([pt], lr=lr, momentum=mom, nesterov=True, weight_decay=wd)
        with BACKENDS["gpu"]() as dev:
            p = torch.from_numpy(p0.copy()).to(dev)
            buf = torch.zeros_like(p)
            for g in grads:
                pt.grad = torch.from_numpy(g.copy())
                opt.step()
                ops.sgd_nesterov_(p, torch.from_numpy(g.copy()).to(dev), buf, lr, mom, wd)
                assert_close(p, pt.detach(), 1e-6, f"sgd-nesterov lr={lr} mom={mom} wd={wd} params")
                assert_close(buf, opt.state[pt]["momentum_buffer"], 1e-6, "momentum buffer")


def test_sgd_nesterov_vs_torch_emu():
    from text_segmentation_image_inpainting_amd import ops
    rng = np.random.default_rng(1601)
    n, lr, mom, wd = 1000, 0.01, 0.9, 1e-3
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([pt], lr=lr, momentum=mom, nesterov=True, weight_decay=wd)
    with BACKENDS["emu"]():
        p = torch.from_numpy(p0.copy())
        buf = torch.zeros_like(p)
        for _ in range(3):
            g = rng.standard_normal(n).astype(np.float32)
            pt.grad = torch.from_numpy(g.copy())
            opt.step()
            ops.sgd_nesterov_(p, torch.from_numpy(g.copy()), buf, lr, mom, wd)
            assert_close(p, pt.detach(), 1e-6, "sgd-nesterov params")
            assert_close(buf, opt.state[pt]["momentum_buffer"], 1e-6, "momentum buffer")


@both_backends
def test_upsample_concat_exact(backend):
    """a6: DoubleUpSample (nearest x2) and the fused up-sample + channel concat are exact copies (forward and backward:
    the backward of nearest x2 is the 2x2 block sum, exact up to fp32 summation order of 4 terms)."""
    from text_segmentation_image_inpainting_amd import ops
    from text_segmentation_image_inpainting_amd.BaseModels import to_nchw, to_nhwc
    with BACKENDS[backend]() as dev:
        for idx, (c1, c2, h, w) in enumerate([(8, 4, 5, 7), (32, 3, 6, 6), (5, 0, 4, 9), (12, 35, 3, 8)]):
            rng = np.random.default_rng(1700 + idx)
            low = torch.from_numpy(rng.standard_normal((2, c1, h, w)).astype(np.float32))
            lo = low.clone().requires_grad_(True)
            up = F.interpolate(lo, scale_factor=2, mode="nearest")
            if c2:
                skip = torch.from_numpy(rng.standard_normal((2, c2, 2 * h, 2 * w)).astype(np.float32))
                so = skip.clone().requires_grad_(True)
                ref = torch.cat([up, so], 1)
            else:
                ref = up
            gy = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
            ref.backward(gy)
            ld = low.to(dev).requires_grad_(True)
            if c2:
                sdv = skip.to(dev).requires_grad_(True)
                out = to_nchw(ops.upcat(to_nhwc(ld), to_nhwc(sdv)))
            else:
                out = to_nchw(ops.upsample2x(to_nhwc(ld)))
            assert torch.equal(out.detach().cpu(), ref.detach()), f"upcat case {idx} forward not exact"
            out.backward(gy.to(dev))
            assert_close(ld.grad, lo.grad, 1e-6, f"upcat case {idx} dlow")
            if c2:
                assert torch.equal(sdv.grad.cpu(), so.grad), f"upcat case {idx} dskip not exact"
        # the module: x and the mask tensor are both up-sampled (models/partial_convolution.py:229-231)
        x = torch.from_numpy(np.random.default_rng(1710).standard_normal((2, 6, 4, 5)).astype(np.float32))
        mk = (torch.from_numpy(np.random.default_rng(1711).uniform(size=(2, 1, 4, 5))) > 0.4).float().expand(-1, 6, -1, -1).contiguous()
        y, m2 = T.DoubleUpSample(scale_factor=2, mode="nearest")((x.to(dev), mk.to(dev)))
        m2 = m2.as_tensor() if hasattr(m2, "as_tensor") else m2
        assert torch.equal(y.cpu(), F.interpolate(x, scale_factor=2, mode="nearest"))
        assert torch.equal(m2.cpu(), F.interpolate(mk, scale_factor=2, mode="nearest"))


@pytest.mark.gpu
def test_textsegament_full_size_properties_gpu():
    """BASELINE config 3 size (TextSegament, 512x512; batch 8 here -- batch 64 is the bench configuration): forward
    is deterministic, an image's eval output does not depend on its batch mates, a training step gives finite
    gradients for every trainable parameter and bit-identical gradients when repeated."""
    with BACKENDS["gpu"]() as dev:
        from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
        x, t = make_seg_batch(8, 512, seed0=300)
        x, t = x.to(dev), t.to(dev)
        torch.manual_seed(0)
        m = T.TextSegament()
        fill_state_dict_(m.state_dict(), seed=47, gain=1.0)
        m = m.to(dev).eval()
        with torch.no_grad():
            y1, y2 = m(x), m(x)
            assert tuple(y1.shape) == (8, 1, 512, 512)
            assert torch.equal(y1, y2) and bool(torch.isfinite(y1).all())
            for i in (0, 5):
                assert_close(m(x[i:i + 1]), y1[i:i + 1], 1e-5, f"batch independence, image {i}")
        crit = T.BinaryFocalLoss(0, 1, 2)
        grads = []
        for _ in range(2):
            m2 = T.TextSegament()
            fill_state_dict_(m2.state_dict(), seed=47, gain=1.0)
            m2 = m2.to(dev).train()
            loss = crit(m2(x), t)
            loss.backward()
            assert bool(torch.isfinite(loss))
            g = [p.grad for p in m2.parameters() if p.requires_grad]
            assert all(v is not None and bool(torch.isfinite(v).all()) for v in g)
            grads.append(torch.cat([v.reshape(-1) for v in g]))
        assert torch.equal(grads[0], grads[1])


@pytest.mark.gpu
def test_gemm_arithmetic_modes_accuracy_gpu(capsys):
    """The split-bf16 matrix products (tsii_set_gemm_products 6 / 8) are fp32-class: their error against an fp64
    reference is within a small factor of the bit-exact f32-MFMA path's (mode 0) own rounding error, for the forward
    (NT), dX (NT) and dW (TN) forms; mode 3 (2 pieces) is the documented 2^-15 class; mode 1 (bf16-rounded operands) is
    held to its definition -- the fp64 product of the rounded operands -- at the same fp32-accumulation bar, in all three forms."""
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    with BACKENDS["gpu"]() as dev:
        L = _lib.lib()
        st = _lib.stream()
        rng = np.random.default_rng(1800)
        M, K, N = 4096, 1024, 256
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        dy = rng.standard_normal((M, N)).astype(np.float32)
        x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
        refs = {"fwd": x64 @ w64.T, "dx": dy64 @ w64, "dw": dy64.T @ x64}
        # mode 1 ("mixed bf16") is DEFINED as the product of the operands rounded to bf16 (nearest even), accumulated in fp32
        xb, wb, dyb = (torch.from_numpy(a).bfloat16().double().numpy() for a in (x, w, dy))
        refs_bf16 = {"fwd": xb @ wb.T, "dx": dyb @ wb, "dw": dyb.T @ xb}
        scales = {"fwd": np.abs(x64) @ np.abs(w64).T, "dx": np.abs(dy64) @ np.abs(w64), "dw": np.abs(dy64).T @ np.abs(x64)}
        xt, wt, dyt = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(dy).to(dev)
        errs = {}
        saved = _lib._GEMM_PRODUCTS
        try:
            for mode in (0, 6, 8, 3, 1):
                _lib.set_gemm_products(mode)          # through the binding: call() re-applies ITS selection on every thread
                assert L.tsii_get_gemm_products() == mode
                y = torch.empty(M, N, device=dev)
                wws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
                call("tsii_pw_fwd", ptr(xt), M, K, ptr(wt), N, None, None, 0, None, None, None, ptr(y), ptr(wws), wws.numel() * 4, st)
                dx = torch.empty(M, K, device=dev)
                wtws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
                call("tsii_pw_bwd_dx", ptr(dyt), M, N, ptr(wt), K, None, None, 0, None, ptr(dx), ptr(wtws), st)
                nb = L.tsii_pw_bwd_dw_ws_bytes(M, N, K)
                ws = torch.empty(nb // 4 + 4, device=dev)
                dw = torch.empty(N, K, device=dev)
                call("tsii_pw_bwd_dw", ptr(dyt), ptr(xt), M, N, K, None, None, None, 0, None, ptr(dw), None, ptr(ws), nb, st)
                torch.cuda.synchronize()
                for name, out in (("fwd", y), ("dx", dx), ("dw", dw)):
                    e = np.abs(out.cpu().numpy().astype(np.float64) - (refs_bf16 if mode == 1 else refs)[name]) / scales[name]
                    errs[(mode, name)] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
        finally:
            _lib.set_gemm_products(saved)
        with capsys.disabled():
            for k, v in errs.items():
                print(f"\n[gemm accuracy] mode {k[0]} {k[1]:3s}: max |err|/sum|a||b| = {v[0]:.3e}  rms = {v[1]:.3e}", end="")
            print()
        for name in ("fwd", "dx", "dw"):
            base_max, base_rms = errs[(0, name)]
            for mode in (6, 8):
                assert errs[(mode, name)][1] <= 2.0 * base_rms + 1e-9, (mode, name, errs[(mode, name)], errs[(0, name)])
                assert errs[(mode, name)][0] <= 3.0 * base_max + 1e-9, (mode, name, errs[(mode, name)], errs[(0, name)])
            assert errs[(3, name)][0] <= 2e-5
            # mode 1 against ITS definition: forward and both backward products within fp32 accumulation error
            assert errs[(1, name)][0] <= 3.0 * base_max + 1e-9, (1, name, errs[(1, name)], errs[(0, name)])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["XceptionTextSegment", "TextSegament"])
def test_demo_end_to_end_gpu(name, tmp_path):
    """n2: examples/demo_segmentation.py end to end on a synthetic tile (Examples/demo_segmentation.py:17-70): a
    checkpoint in the public format (a plain ``state_dict`` saved with ``torch.save``; here with one foreign key and one
    wrongly-shaped entry, as after a head change) goes through the tolerant loader; EvaluateSet pads / normalises; the
    net runs on the GPU; sigmoid > 0.5, 3x3 max-pool, un-pad + resize give the mask.  Checked against the CPU oracle
    run on the very same network input (pixels within rounding distance of the threshold may flip: <= 0.5 %)."""
    import importlib.util
    from PIL import Image
    from text_segmentation_image_inpainting_amd.Dataloader import EvaluateSet
    from text_segmentation_image_inpainting_amd.synthetic import manga_tile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("demo", os.path.join(root, "examples", "demo_segmentation.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    keys = json.load(open(os.path.join(GOLD, "seg_state_dict_keys.json")))[name]
    sd = make_state_dict([(k, s) for k, s in keys], seed=51, gain=1.0)
    ckpt = dict(sd)
    ckpt["classifier.weight"] = torch.zeros(3)                                   # not in the model: reported, skipped
    ckpt[keys[0][0]] = torch.zeros(5, 7)                                         # incompatible shape: reported, skipped
    torch.save(ckpt, tmp_path / "ckpt.pt")
    tile = (manga_tile(200, np.random.default_rng(3)).transpose(1, 2, 0) * 255).astype(np.uint8)
    Image.fromarray(tile).save(tmp_path / "tile.png")
    with BACKENDS["gpu"]() as dev:
        torch.manual_seed(0)
        model = getattr(T, name)()
        fill_state_dict_(model.state_dict(), seed=51, gain=1.0)               # what the skipped entry keeps
        unknown, failed = model.load_state_dict(torch.load(tmp_path / "ckpt.pt", map_location="cpu"))
        assert unknown == ["classifier.weight"] and failed == [keys[0][0]]
        model = model.to(dev).eval()
        evalset = EvaluateSet(mean=[0.4935, 0.4563, 0.4544], std=[0.3769, 0.3615, 0.3566], img_folder=str(tmp_path), resize=256)
        item = evalset[0]
        mask = demo.process(model, item, dev)
        (img, origin, unpadder), fname = item
        assert os.path.exists(fname + "_mask.jpg") and os.path.exists(fname + "_contour.jpg")
        assert tuple(mask.shape[-2:]) == tuple(origin.shape[-2:]) and set(torch.unique(mask).tolist()) <= {0.0, 1.0}
        with torch.no_grad():
            ref_logits = S.SEG_MODELS[name](sd, img, training=False)
        ref = unpadder(demo.max_pool3x3_binary((ref_logits > 0)).byte()).float()
        assert float((ref != mask).float().mean()) <= 5e-3


@pytest.mark.gpu
def test_mixed_bf16_products_mode_gpu(capsys):
    """BASELINE config 5's arithmetic ("mixed bf16"): gemm products = 1 rounds the operands of every matrix product to bf16
    (one MFMA product, fp32 accumulation; storage, BatchNorm, stencils stay fp32).  There is no reference code for it
    (models/ACNN.py is un-importable), so the yardsticks are
      * kernel level (test_gemm_arithmetic_modes_accuracy_gpu): forward / dX / dW equal the fp64 product of the bf16-ROUNDED
        operands to fp32 accumulation accuracy -- the mode computes exactly what it says, backward included;
      * block level, here: a kink-free RFB block (train-mode BatchNorm chains four convolutions deep, every kernel family of
        the segmentation path) forward + dX + every parameter gradient within 5e-2 of the fp64 oracle (measured: 3.6e-2 worst, a BatchNorm bias);
      * net level, here: XceptionTextSegment against the reference's fp32 / fp64 fixture -- eval forward (3e-2), train-mode
        forward with batch statistics (5e-2), focal loss (2e-2), and the gradients of the decoder head (0.1).  The encoder's
        gradients are bounded at what the arithmetic allows, not at the fp32 bar: this ~100-layer train-mode BatchNorm net on
        8x8 maps amplifies operand rounding by ~1e4 (its fp32 runs already differ by 7e-4 from fp64, SURVEY.md F11;
        tests/test_parity_seg.py) and every rounded activation near a LeakyReLU kink flips a derivative, so bf16 operand
        rounding (4e-3) shows up at median 0.25, worst 0.46-0.49 of a tensor's largest entry (rounds 3 / 4, measured): every
        recorded tensor must stay within 0.8, the median within 0.4, and every tensor must keep the DIRECTION of the fp64
        gradient (cosine >= 0.5; measured >= 0.8) -- a wrong kernel gives O(1) errors with no such correlation.  (The kernels
        themselves are pinned by the two levels above; tests/test_bf16_storage.py does the same for bf16 STORAGE, with a smooth
        variant of the net in which every gradient tensor meets a 1e-2-class bar.)"""
    from text_segmentation_image_inpainting_amd import _lib
    G = np.load(os.path.join(GOLD, "xceptiontextsegment_64.npz"))
    with BACKENDS["gpu"]() as dev:
        saved = _lib._GEMM_PRODUCTS
        try:
            m = T.XceptionTextSegment()
            fill_state_dict_(m.state_dict(), seed=41, gain=1.0)
            m = m.to(dev).eval()
            x, t = torch.from_numpy(G["x"]).to(dev), torch.from_numpy(G["t"]).to(dev)
            errs = {}
            for mode in (6, 3, 1):
                _lib.set_gemm_products(mode)          # kept by the package and applied on every calling thread (autograd's too)
                with torch.no_grad():
                    y = m(x)
                errs[mode] = float((y.cpu().double() - torch.from_numpy(G["y_eval_f64"]).double()).abs().max() / np.abs(G["y_eval_f64"]).max())
            # train mode, forward + backward in the bf16-operand arithmetic
            m.train()
            y = m(x)
            e_train = rel_err(y, G["y_train"])
            loss = T.BinaryFocalLoss(0, 1, 2)(y, t)
            e_loss = abs(loss.item() - float(G["loss"])) / abs(float(G["loss"]))
            loss.backward()
            params = dict(m.named_parameters())
            gmax = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith("grad64."))
            gerr = sorted((rel_err(params[k[7:]].grad, G[k].astype(np.float32), 1e-3 * gmax), k[7:]) for k in G.files if k.startswith("grad64."))
            head = [(e, k) for e, k in gerr if k.startswith(("out_conv", "feature_4x_conv"))]
            cosines = []
            for k in G.files:
                if k.startswith("grad64."):
                    a_, b_ = params[k[7:]].grad.detach().cpu().double().reshape(-1), torch.from_numpy(G[k]).double().reshape(-1)
                    if float(b_.abs().max()) > 1e-3 * gmax:
                        cosines.append((float((a_ * b_).sum() / (a_.norm() * b_.norm() + 1e-300)), k[7:]))
            e_blk = _rfb_case(dev, 64, 64, 32, seed=1401, smooth=True, coarse_bar=5e-2)
        finally:
            _lib.set_gemm_products(saved)
        assert _lib.get_gemm_products() == (6 if saved is None else saved)
        with capsys.disabled():
            print("\n[mixed bf16] XceptionTextSegment 64x64 eval, max-normalised error vs the reference's fp64 run: " +
                  ", ".join(f"products={k}: {v:.2e}" for k, v in errs.items()))
            print(f"[mixed bf16] train mode, products=1: output {e_train:.2e}, focal loss {e_loss:.2e}; gradients vs fp64: decoder head worst "
                  f"{max(head)[0]:.2e} over {len(head)} tensors; all {len(gerr)} recorded tensors median {gerr[len(gerr) // 2][0]:.2e}, worst "
                  f"{gerr[-1][0]:.2e} ({gerr[-1][1]})")
            print(f"[mixed bf16] kink-free RFB 64->32 on 64x64, train mode, products=1: forward {e_blk[0]:.2e}, dX / parameter gradients {e_blk[1]:.2e}")
        assert errs[6] <= 1e-3 and errs[3] <= 1e-3 and errs[1] <= 3e-2
        assert e_train <= 5e-2 and e_loss <= 2e-2
        assert len(head) >= 4 and max(head)[0] <= 0.1
        with capsys.disabled():
            print(f"[mixed bf16] direction of the recorded gradients vs fp64: worst cosine {min(cosines)[0]:.3f} ({min(cosines)[1]})")
        assert gerr[-1][0] <= 0.8 and gerr[len(gerr) // 2][0] <= 0.4, (gerr[-1], gerr[len(gerr) // 2])
        assert min(cosines)[0] >= 0.5, min(cosines)


@both_backends
@pytest.mark.parametrize("d", [2, 4, 8])
def test_fused_dilated_blocks_taps_in_range(backend, d):
    """The BatchNorm-fused forms of the dilated marching strips (K6b BatchNorm-on-load + statistics epilogue, K6c reductions
    in the dX strip, BatchNorm-on-load dW; dilation 2 / 4 / 8) on maps wider than 2d, i.e. with every tap in range -- the
    fixtures reach these dilations only on 8x8 maps (centre tap).  ``InvertedResidual`` (segmentation path) and
    ``PartialInvertedResidual`` (ImageFill's dilated stages, image_inpainting.py:33-37) in train mode: output, dX, every
    parameter gradient, running statistics vs the oracle."""
    from oracle.filler import seeded_input
    from text_segmentation_image_inpainting_amd import ops
    H, W, c, t = 2 * d + 21, 2 * d + 26, 16, 2
    act = torch.nn.LeakyReLU(0.3)
    rng = np.random.default_rng(40 + d)
    calls = []
    real = ops.call
    with BACKENDS[backend]() as dev:
        ops.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        try:
            # ---- InvertedResidual: 1x1 + BN + act -> dw 3x3 (dilation d) + BN + act -> 1x1 + BN, residual
            m = T.InvertedResidual(c, c, 1, t, d, activation=act)
            keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
            fill_state_dict_(m.state_dict(), seed=40 + d)
            sd = make_state_dict(keys, seed=40 + d)
            names = [k for k, _ in m.named_parameters()]
            for k in names:
                sd[k].requires_grad_(True)
            x = torch.from_numpy(rng.standard_normal((2, c, H, W)).astype(np.float32))
            xo = x.clone().requires_grad_(True)
            yo = S.inverted_residual(sd, "", xo, c, c, 1, t, d, act, False, True)
            gy = torch.from_numpy(rng.standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            m = m.to(dev).train()
            xd = x.to(dev).requires_grad_(True)
            y = m(xd)
            assert_close(y, yo, TOL, f"IR d={d} y")
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, f"IR d={d} dx")
            for k, p in m.named_parameters():
                assert_close(p.grad, sd[k].grad, 2e-3, f"IR d={d} grad {k}", floor=1e-6)
            for k, v in m.state_dict().items():
                if "running" in k:
                    assert_close(v, sd[k], TOL, f"IR d={d} {k}")
            # ---- PartialInvertedResidual with a hole mask
            pm = T.PartialInvertedResidual(c, c, 3, 1, d, d, t, BN=True, activation=act, use_1_conv=True, same_holes=True)
            keys = [(k, tuple(v.shape)) for k, v in pm.state_dict().items()]
            fill_state_dict_(pm.state_dict(), seed=50 + d)
            sd = make_state_dict(keys, seed=50 + d)
            names = [k for k, p in pm.named_parameters() if p.requires_grad]
            for k in names:
                sd[k].requires_grad_(True)
            x, mask = seeded_input(2, c, H, W, seed=50 + d, hole_frac=0.15)
            xo = x.clone().requires_grad_(True)
            yo, nmo = O.partial_inverted_residual(sd, "", xo, mask, c, c, 3, 1, d, d, t, O.leaky(0.3), True, False, True, True)
            yo.backward(gy)
            pm = pm.to(dev).train()
            xd = x.to(dev).requires_grad_(True)
            y, nm = pm((xd, mask.to(dev)))
            assert_close(y, yo, TOL, f"PIR d={d} y")
            assert np.array_equal((nm.as_tensor() if hasattr(nm, "as_tensor") else nm).detach().cpu().numpy(), nmo.numpy())
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, f"PIR d={d} dx")
            for k in names:
                assert_close(dict(pm.named_parameters())[k].grad, sd[k].grad, 2e-3, f"PIR d={d} grad {k}", floor=1e-6)
        finally:
            ops.call = real
    # the fused entry points really ran (not the materialised fall-back); at dilation 2 / 4 the ring kernel's dX pass also takes the
    # weight gradient (K6d), at dilation 8 these maps belong to the row-phase kernel's geometry, which has no such form
    bwd = ("tsii_dw_bwd_dxdw_bn",) if d in (2, 4) else ("tsii_dw_bwd_dx_bn", "tsii_dw_bwd_dw_bn")
    for name in ("tsii_dw_fwd_bn", "tsii_bn_act_bwd_pre") + bwd:
        assert name in calls, (name, sorted(set(calls)))


@both_backends
def test_fused_stride2_block_multi_strip(backend):
    """K6c in the stride-2 dX strip kernel (the BatchNorm-backward reductions of the expand BatchNorm taken while dX is
    written) on an odd-sized map spanning several strips and row chunks: ``PartialInvertedResidual`` stride 2 in train mode
    vs the oracle (the reference-generated fixture of this flavour is 12x12)."""
    from oracle.filler import seeded_input
    from text_segmentation_image_inpainting_amd import ops
    c, t, H, W = 16, 4, 45, 53
    act = torch.nn.LeakyReLU(0.3)
    calls = []
    real = ops.call
    with BACKENDS[backend]() as dev:
        ops.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        try:
            pm = T.PartialInvertedResidual(c, 2 * c, 3, 2, 1, 1, t, BN=True, activation=act, use_1_conv=True, same_holes=True)
            keys = [(k, tuple(v.shape)) for k, v in pm.state_dict().items()]
            fill_state_dict_(pm.state_dict(), seed=61)
            sd = make_state_dict(keys, seed=61)
            names = [k for k, p in pm.named_parameters() if p.requires_grad]
            for k in names:
                sd[k].requires_grad_(True)
            x, mask = seeded_input(3, c, H, W, seed=61, hole_frac=0.15)
            xo = x.clone().requires_grad_(True)
            yo, nmo = O.partial_inverted_residual(sd, "", xo, mask, c, 2 * c, 3, 2, 1, 1, t, O.leaky(0.3), True, False, True, True)
            gy = torch.from_numpy(np.random.default_rng(62).standard_normal(tuple(yo.shape)).astype(np.float32))
            yo.backward(gy)
            pm = pm.to(dev).train()
            xd = x.to(dev).requires_grad_(True)
            y, nm = pm((xd, mask.to(dev)))
            assert_close(y, yo, TOL, "PIR s2 y")
            assert np.array_equal((nm.as_tensor() if hasattr(nm, "as_tensor") else nm).detach().cpu().numpy(), nmo.numpy())
            y.backward(gy.to(dev))
            assert_close(xd.grad, xo.grad, TOL, "PIR s2 dx")
            for k in names:
                assert_close(dict(pm.named_parameters())[k].grad, sd[k].grad, 2e-3, f"PIR s2 grad {k}", floor=1e-6)
        finally:
            ops.call = real
    # (K6d / K6e: the stride-2 dX + K6c pass also takes the weight gradient and applies the following BatchNorm's backward on load, so that
    # BatchNorm only reduces; the expand BatchNorm keeps its stand-alone apply pass)
    assert "tsii_dw_bwd_dxdw_bn2" in calls and "tsii_bn_bwd_reduce" in calls and "tsii_dw_bwd_dw_bn" not in calls and calls.count("tsii_bn_act_bwd_pre") >= 1, sorted(set(calls))

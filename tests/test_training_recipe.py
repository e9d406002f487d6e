"""Row n3: the reference's training recipes as ONE loop each -- InpaintingLoss through a frozen MobileNetV2
extractor + SGD-Nesterov + weight decay + cyclical learning rate for the inpainting net; BinaryFocalLoss with the two-stage
encoder freezing for the segmentation net (recipes.py) -- against the same loops written with the oracle's functional
networks / losses, ``torch.optim.SGD(nesterov=True)`` and ``torch.optim.lr_scheduler.CyclicLR`` on the CPU.

GPU: ImageFill 64^2 / TextSegament 64^2 (the golden-fixture sizes).  Emulator (CPU suite): the same loops on a three-block
partial-conv net / a width-0.25 TextSegament at 32^2 so the suite stays in minutes."""
import json
import os

import numpy as np
import pytest
import torch
from torch import nn

import text_segmentation_image_inpainting_amd as T
from oracle import pconv_oracle as O
from oracle import seg_oracle as S
from oracle.filler import fill_state_dict_, make_state_dict, seeded_input
from tests.backends import BACKENDS

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EPS32 = float(np.finfo(np.float32).eps)


def _torch_loop(params, base_lr, max_lr, step_size, wd):
    opt = torch.optim.SGD(params, lr=base_lr, momentum=0.9, nesterov=True, weight_decay=wd)
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=base_lr, max_lr=max_lr, step_size_up=step_size, mode="triangular",
                                              cycle_momentum=False)
    return opt, sched


def _check_losses(losses, ref):
    """Step 0 sees identical parameters: tight.  Later steps see parameters that already carry the (noise-limited, SURVEY.md
    F11) gradient error of the earlier updates, which moves the loss by a fraction of what the update itself moved it:
    bound the error by 5 % of the reference's loss change over that step."""
    assert abs(losses[0] - ref[0]) <= 2e-5 * abs(ref[0]), (losses, ref)
    for i in range(1, len(ref)):
        assert abs(losses[i] - ref[i]) <= 5e-2 * abs(ref[i] - ref[i - 1]) + 2e-5 * abs(ref[i]), (i, losses, ref)


def _update_error(p_hip, p_ref, p_start):
    """|p_hip - p_ref| relative to the size of the update the optimizer made to this tensor, after allowing 4 ulp of the
    parameter itself (at lr 1e-4 an update is a few dozen ulp of a weight: the comparison must not be made on the
    quantised difference p_end - p_start)."""
    upd = float((p_ref - p_start).abs().max())
    slack = 4 * EPS32 * float(p_ref.abs().max())
    return max(0.0, float((p_hip - p_ref).abs().max()) - slack) / max(upd, 1e-12)


class TinyFill(nn.Module):
    """stem partial conv (bias, LeakyReLU) -> PartialInvertedResidual (three BatchNorms, residual) -> 3-channel head"""

    def __init__(self):
        super().__init__()
        act = nn.LeakyReLU(0.3)
        self.stem = T.partial_convolution_block(3, 8, 3, 1, 1, 1, bias=True, BN=False, activation=act)
        self.body = T.PartialInvertedResidual(8, 8, 3, 1, 1, 1, 2, BN=True, activation=act, use_1_conv=True, same_holes=True)
        self.head = T.partial_convolution_block(8, 3, 3, 1, 1, 1, bias=True, BN=False, activation=None)

    def forward(self, args):
        return self.head(self.body(self.stem(args)))[0]


def _tiny_fill_oracle(sd, x, mask):
    act = O.leaky(0.3)
    h, m = O.pconv_block(sd, "stem.", x, mask, 1, 1, 1, 1, BN=False, act=act)
    h, m = O.partial_inverted_residual(sd, "body.", h, m, 8, 8, 3, 1, 1, 1, 2, act, True, False, True, True)
    return O.pconv_block(sd, "head.", h, m, 1, 1, 1, 1, BN=False, act=None)[0]


def _inpainting_recipe_case(backend, make_model, oracle_fwd, key_shapes, trainable, size, tol):
    from text_segmentation_image_inpainting_amd.recipes import InpaintingRecipe
    batch, steps = 2, 3
    cfg = dict(base_lr=1e-4, max_lr=1e-2, step_size=2)
    _, mask = seeded_input(batch, 3, size, size, seed=5, hole_frac=0.15)
    clean = torch.from_numpy(np.random.default_rng(6).uniform(0, 1, (batch, 3, size, size)).astype(np.float32))
    corrupted = clean * mask
    # ---- oracle loop (stock torch, CPU)
    sd = make_state_dict(key_shapes, seed=3)
    for k in trainable:
        sd[k].requires_grad_(True)
    crit_keys = T.InpaintingLoss(T.MobileNetV2(width_mult=1), feature_range=3).state_dict()
    ext_sd = make_state_dict([(k, tuple(v.shape)) for k, v in crit_keys.items()], seed=77)
    opt, sched = _torch_loop([sd[k] for k in trainable], cfg["base_lr"], cfg["max_lr"], cfg["step_size"], 1e-4)
    start = {k: sd[k].detach().clone() for k in trainable}
    ref_losses, ref_lrs = [], []
    for _ in range(steps):
        opt.zero_grad()
        loss = S.inpainting_loss(ext_sd, corrupted, mask, oracle_fwd(sd, corrupted, mask), clean)
        loss.backward()
        ref_lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        ref_losses.append(float(loss.detach()))
    # ---- the recipe on the HIP path
    with BACKENDS[backend]() as dev:
        model = make_model()
        fill_state_dict_(model.state_dict(), seed=3)
        rec = InpaintingRecipe(model.to(dev).train(), T.MobileNetV2(width_mult=1), feature_range=3, weight_decay=1e-4, **cfg)
        fill_state_dict_(rec.criterion.state_dict(), seed=77)
        rec.to(dev)
        losses, lrs = [], []
        for _ in range(steps):
            losses.append(float(rec.step(corrupted.to(dev), mask.to(dev), clean.to(dev))))
            lrs.append(rec.lr)
        assert all(not p.requires_grad for p in rec.criterion.parameters())
        assert np.allclose(lrs, ref_lrs, rtol=1e-12) and lrs[0] == cfg["base_lr"] and lrs[2] == cfg["max_lr"]   # 1e-4, 5.05e-3, 1e-2
        _check_losses(losses, ref_losses)
        params = dict(model.named_parameters())
        assert sorted(k for k, p in params.items() if p.requires_grad) == sorted(trainable)
        # per tensor (cancellation-heavy ones -- BatchNorm gammas under the Gram-matrix terms -- carry the noise-limited
        # gradient error of train-mode BatchNorm nets almost undamped: loose bound), and over the whole update vector (tight)
        worst = max((_update_error(params[k].detach().cpu(), sd[k].detach(), start[k]), k) for k in trainable)
        du_hip = torch.cat([(params[k].detach().cpu() - start[k]).reshape(-1).double() for k in trainable])
        du_ref = torch.cat([(sd[k].detach() - start[k]).reshape(-1).double() for k in trainable])
        rel = float((du_hip - du_ref).norm() / du_ref.norm())
        print(f"[recipe {backend}] update vector: relative L2 error {rel:.3e}; worst tensor {worst[1]} {worst[0]:.3e}")
        assert rel <= tol and worst[0] <= 10 * tol, (rel, worst)
        # the extractor stays in train mode in the reference: its running statistics move, and match the oracle's
        k0 = "feature_encoder.layers.0.1.0.running_mean"
        assert float((rec.criterion.state_dict()[k0].cpu() - ext_sd[k0]).abs().max()) <= 2e-3 * float(ext_sd[k0].abs().max())   # fed by the (slightly diverged) outputs


def test_inpainting_recipe_tiny_net_emu():
    probe = TinyFill()
    key_shapes = [(k, tuple(v.shape)) for k, v in probe.state_dict().items()]
    trainable = [k for k, p in probe.named_parameters() if p.requires_grad]
    _inpainting_recipe_case("emu", TinyFill, _tiny_fill_oracle, key_shapes, trainable, size=24, tol=2e-2)


@pytest.mark.gpu
def test_inpainting_recipe_imagefill_gpu():
    keys = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))
    _inpainting_recipe_case("gpu", T.ImageFill, lambda sd, x, m: O.image_fill(sd, x, m, training=True),
                            [(k, s) for k, s in keys["ImageFill"]], list(keys["ImageFill.trainable"]), size=64, tol=5e-2)


def _segmentation_recipe_case(backend, width_mult, x, t, tol):
    from text_segmentation_image_inpainting_amd.recipes import SegmentationRecipe
    cfg = dict(base_lr=1e-4, max_lr=4e-4, step_size=2)
    probe = T.TextSegament(width_mult=width_mult)
    keys = [(k, tuple(v.shape)) for k, v in probe.state_dict().items()]
    all_params = [k for k, _ in probe.named_parameters()]
    stage1 = [k for k in all_params if not k.startswith("encoder.")]
    # ---- oracle: stage 1 trains everything outside `encoder.`, stage 2 everything (fresh optimizer + schedule)
    sd = make_state_dict(keys, seed=41, gain=1.0)
    start = {k: sd[k].detach().clone() for k in all_params}
    ref_losses = []
    for names, nsteps in ((stage1, 2), (all_params, 1)):
        for k in all_params:
            sd[k].requires_grad_(k in names)
        opt, sched = _torch_loop([sd[k] for k in names], cfg["base_lr"], cfg["max_lr"], cfg["step_size"], 1e-3)
        for _ in range(nsteps):
            opt.zero_grad()
            loss = S.binary_focal_loss(S.text_segament(sd, x, training=True, width_mult=width_mult), t, 0.0, 1.0, 2.0)
            loss.backward()
            opt.step()
            sched.step()
            ref_losses.append(float(loss.detach()))
        if names is stage1:
            mid = {k: sd[k].detach().clone() for k in all_params}
    with BACKENDS[backend]() as dev:
        net = T.TextSegament(width_mult=width_mult)
        fill_state_dict_(net.state_dict(), seed=41, gain=1.0)
        net = net.to(dev).train()
        rec = SegmentationRecipe(net, free_last_blocks=0, weight_decay=1e-3, **cfg)
        enc0 = {k: v.detach().clone() for k, v in net.encoder.named_parameters()}
        bn0 = net.state_dict()["encoder.features.0.1.0.running_mean"].clone()
        assert all(not p.requires_grad for p in net.encoder.parameters()) and len(rec.trainer.params) == len(stage1)
        losses = [float(rec.step(x.to(dev), t.to(dev))) for _ in range(2)]
        # stage 1: encoder weights bit-identical, its BatchNorm statistics still moving (train mode), decoder updated
        assert all(torch.equal(v, dict(net.encoder.named_parameters())[k].detach()) for k, v in enc0.items())
        assert not torch.equal(bn0, net.state_dict()["encoder.features.0.1.0.running_mean"])
        params = dict(net.named_parameters())
        worst1 = max((_update_error(params[k].detach().cpu(), mid[k], start[k]), k) for k in stage1)
        rec.unfreeze()
        assert rec.stage == 2 and len(rec.trainer.params) == len(all_params) and rec.lr == cfg["base_lr"]
        losses.append(float(rec.step(x.to(dev), t.to(dev))))
        params = dict(net.named_parameters())
        worst2 = max((_update_error(params[k].detach().cpu(), sd[k].detach(), mid[k]), k) for k in all_params)
        _check_losses(losses, ref_losses)
        assert worst1[0] <= tol and worst2[0] <= tol, (worst1, worst2)
        assert any(not torch.equal(v, params["encoder." + k].detach()) for k, v in enc0.items())     # stage 2 trains the encoder


def test_segmentation_two_stage_recipe_emu():
    from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
    x, t = make_seg_batch(2, 32, seed0=3)
    _segmentation_recipe_case("emu", 0.25, x, t, tol=5e-2)


@pytest.mark.gpu
def test_segmentation_two_stage_recipe_gpu():
    G = np.load(os.path.join(GOLD, "textsegament_64.npz"))
    _segmentation_recipe_case("gpu", 2, torch.from_numpy(G["x"]), torch.from_numpy(G["t"]), tol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--model", "XceptionTextSegment", "--size", "128", "--batch", "2", "--products", "1"],
                                   ["--model", "TextSegament", "--size", "128", "--batch", "2", "--pixel-shuffle", "--checkpoint"],
                                   ["--model", "ImageFill", "--size", "128", "--batch", "2"]])
def test_bench_line_contract_gpu(extra):
    """bench.py end to end on small shapes: ONE JSON line with the contract's keys, for the headline workload and for the
    secondary configs (cfg 3 / cfg 5 flags)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "kernel_classes", "forward_only", "f32_mfma_mode", "split3_mode", "comm"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["gemm_products"] == (1 if "--products" in extra else 6)
    assert ("headline" not in d["metric"]) == (extra[1] == "ImageFill")

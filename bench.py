#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): imgs/s of the partial-conv inpainting training step
(ImageFill, 512x512, batch 32 per GPU, train-mode BN, fwd + bwd + gradient all-reduce + fused
SGD update) on N MI355X of one node, with the roofline of the dominant kernel class, the
per-class rooflines and the CPU baseline (the oracle timed on the host cores) in the same JSON line.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3    # MI355X f32-input MFMA = fp32 vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide)
PEAK_HBM_TBS = 8.0          # HBM3E spec
# ImageFill 512^2 forward = 58.8 GFLOP and 2934 MB (train-mode BN) per image; fwd+bwd = 3x
ALG_GFLOP_PER_IMG = 3 * 58.8
ALG_GB_PER_IMG = 3 * 2.934
# the same table for every net / size it lists: (GFLOP forward, GB forward in train mode at 4 bytes per element) per image
WHOLE_FWD = {("ImageFill", 512): (58.8, 2.934), ("ImageFillOrigin", 512): (75.9, 0.3676), ("ImageFillOriginV2", 512): (78.8, 0.5664),
             ("TextSegament", 256): (22.7, 0.950), ("TextSegament", 512): (90.6, 3.801), ("TextSegament", 1024): (362.5, 15.205),
             ("XceptionTextSegment", 256): (37.2, 0.957), ("XceptionTextSegment", 512): (148.7, 3.828), ("XceptionTextSegment", 1024): (594.8, 15.314)}
PMC_SUMMARY = "r06ab_pmc_hbm_traffic_bs32.json"   # tools/collect_profiles.sh -> tools/pmc_summary.py


PMC_SUMMARY_CFG5 = "r05h_pmc_hbm_traffic_cfg5_bf16storage.json"   # the same two passes around the cfg 5 line in bf16 storage


def csrc_sha(all_files=False):
    """Fingerprint of the kernel sources of a profiled configuration; the committed PMC summary carries the one it was measured at.
    ImageFill in fp32 storage (the headline): every file but the bf16-storage kernels, which it never launches; all_files: everything
    (cfg 5 in bf16 storage launches kernels of both families)."""
    d = os.path.join(ROOT, "text_segmentation_image_inpainting_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and (all_files or not f.startswith("bf16_")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ---- algorithmic work per C-ABI call (SURVEY.md 8(d) traffic model), decoded from the call's scalar arguments ----------
# class -> what bounds it; every entry: (class, lambda scalar_args -> (algorithmic bytes, MACs))
def _pw(a):          # (m, k, n, ...): read [m,k] once, write [m,n] once
    m, k, n = a[0], a[1], a[2]
    return 4.0 * m * (k + n), float(m) * k * n


def _dw(a, passes):  # (n, h, w, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, ...)
    n, h, w, c, ho, wo = a[0], a[1], a[2], a[3], a[12], a[13]
    return 4.0 * n * c * (h * w * passes[0] + ho * wo * passes[1]), float(n) * ho * wo * c * a[4] * a[5]


def _bn(a, passes):  # (m, c, ...)
    return 4.0 * a[0] * a[1] * passes, 0.0


def _dense(a, rd_in, rd_out, wr_in, wr_out):
    # (n, h, w, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, ...) after the leading int of the _fwd forms is dropped
    n, h, w, cin, cout, kh, kw, ho, wo = a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[13], a[14]
    return (4.0 * n * (h * w * cin * (rd_in + wr_in) + ho * wo * cout * (rd_out + wr_out)),
            float(n) * ho * wo * cout * cin * kh * kw)


def _upcat(a, bwd):  # (n, h, w, c1, c2): low [n,h,w,c1] + skip [n,2h,2w,c2] <-> out [n,2h,2w,c1+c2]
    n, h, w, c1, c2 = a[:5]
    return 4.0 * n * h * w * (c1 + 4 * c2 + 4 * (c1 + c2)), 0.0


def _headcat(a, io_in, io_out):
    c1, c2, n, h, w, cout = a[:6]
    return 4.0 * n * (io_in * (h * w // 4 * c1 + h * w * c2) + io_out * h * w * cout), float(n) * h * w * cout * (c1 + c2) * 9


CALLS = {
    "tsii_pw_fwd": ("gemm_nt", _pw), "tsii_pw_fwd_bn": ("gemm_nt", _pw),
    "tsii_pw_bwd_dx": ("gemm_nt", lambda a: _pw((a[0], a[1], a[2]))),
    "tsii_pw_bwd_dx_bn": ("gemm_nt", lambda a: (_pw((a[0], a[1], a[2]))[0] + 4.0 * a[0] * a[2], _pw((a[0], a[1], a[2]))[1])),
    "tsii_pw_bwd_dw": ("gemm_tn", lambda a: _pw((a[0], a[1], a[2]))), "tsii_pw_bwd_dw_bn": ("gemm_tn", lambda a: _pw((a[0], a[1], a[2]))),
    "tsii_dw_fwd": ("dw_stencil", lambda a: _dw(a, (1, 1))), "tsii_dw_fwd_bn": ("dw_stencil", lambda a: _dw(a, (1, 1))),
    "tsii_dw_bwd_dx": ("dw_stencil", lambda a: _dw(a, (1, 1))),
    "tsii_dw_bwd_dx_bn": ("dw_stencil", lambda a: _dw(a, (2, 1))),      # + the raw BatchNorm input read alongside (K6c)
    "tsii_dw_bwd_dxdw_bn": ("dw_stencil", lambda a: _dw(a, (2, 1))),    # K6d: the same pass also leaves the weight gradient ([C][9]: no traffic to speak of)
    # K6e: (bn2_act, bn2_slope, n, h, ...): reads the gradient and the raw input of the FOLLOWING BatchNorm on the output grid (its apply
    # pass rides here), the layer's raw input on the input grid; writes dX
    "tsii_dw_bwd_dxdw_bn2": ("dw_stencil", lambda a: _dw(a[2:], (2, 2))),
    "tsii_dw_bwd_dw": ("dw_stencil", lambda a: _dw(a, (1, 1))), "tsii_dw_bwd_dw_bn": ("dw_stencil", lambda a: _dw(a, (1, 1))),
    "tsii_bn_act_fwd": ("bn_act", lambda a: _bn(a, 2)), "tsii_bn_stats": ("bn_act", lambda a: _bn(a, 1)),
    "tsii_bn_act_bwd": ("bn_bwd", lambda a: _bn(a, 3)), "tsii_bn_act_bwd_pre": ("bn_bwd", lambda a: _bn(a, 3)),
    "tsii_bn_act_bwd_pre_pool": ("bn_bwd", lambda a: _bn(a, 3.25)),       # + the pooled addend gradient [m/4, c] (K7b)
    "tsii_act_fwd": ("bn_act", lambda a: (8.0 * a[0], 0.0)), "tsii_act_bwd": ("bn_act", lambda a: (12.0 * a[0], 0.0)),
    "tsii_dense_fwd": ("dense_conv", lambda a: _dense(a[1:], 1, 0, 0, 1)), "tsii_dense_fwd_bn": ("dense_conv", lambda a: _dense(a[1:], 1, 0, 0, 1)),
    "tsii_dense_bwd_dx": ("dense_conv", lambda a: _dense(a[1:], 0, 1, 1, 0)),
    "tsii_dense_bwd_dw": ("dense_conv", lambda a: _dense(a[1:], 1, 1, 0, 0)),
    # K4c head over the virtual concatenation: (c1, c2, n, h, w, cout, ...): low [n,h/2,w/2,c1] + skip [n,h,w,c2] <-> y [n,h,w,cout]
    "tsii_head_cat_fwd": ("dense_conv", lambda a: _headcat(a, 1, 1)), "tsii_head_cat_bwd_dx": ("dense_conv", lambda a: _headcat(a, 1, 1)),
    "tsii_head_cat_bwd_dw": ("dense_conv", lambda a: _headcat(a, 1, 1)),
    "tsii_head_cat_fwd_low": ("dense_conv", lambda a: _headcat(a, 1, 1)), "tsii_head_cat_bwd_dw_low": ("dense_conv", lambda a: _headcat(a, 1, 1)),
    "tsii_head_cat_bwd_low": ("dense_conv", lambda a: _headcat(a, 1, 1)),      # dW + d low in one pass
    "tsii_upcat_fwd": ("upcat", lambda a: _upcat(a, False)), "tsii_upcat_bwd": ("upcat", lambda a: _upcat(a, True)),
    # K7b: the high-resolution half of a 1x1 conv over cat(up2(low), skip): (m, k, n, ..): reads [m,k] and the [m/4,n] addend, writes [m,n]
    "tsii_pw_fwd_up": ("gemm_nt", lambda a: (4.0 * a[0] * (a[1] + a[2]) + 1.0 * a[0] * a[2], float(a[0]) * a[1] * a[2])),
    # its addend's gradient: (n, h, w, c) of the LOW grid: reads [n,2h,2w,c], writes [n,h,w,c]
    "tsii_pool2x2_scaled": ("upcat", lambda a: (4.0 * a[0] * a[1] * a[2] * a[3] * 5, 0.0)),
}


def _half(fn):
    """the same traffic model at 2 bytes per element (bf16 activation storage: tsii_bf16_* entry points)"""
    def g(a):
        by, macs = fn(a)
        return by / 2.0, macs
    return g


CALLS.update({
    "tsii_bf16_pw_fwd": ("gemm_nt", _half(_pw)),
    "tsii_bf16_pw_bwd_dx": ("gemm_nt", _half(lambda a: (_pw((a[0], a[1], a[2]))[0] + (4.0 * a[0] * a[2] if a[3] else 0.0), _pw((a[0], a[1], a[2]))[1]))),
    "tsii_bf16_pw_bwd_dw": ("gemm_tn", _half(lambda a: _pw((a[0], a[1], a[2])))),
    "tsii_bf16_dw_fwd": ("dw_stencil", _half(lambda a: _dw(a, (1, 1)))),
    "tsii_bf16_dw_bwd_dx": ("dw_stencil", _half(lambda a: _dw(a, (2 if a[14] else 1, 1)))),     # eps != 0: K6c reads the raw BatchNorm input too
    "tsii_bf16_dw_bwd_dw": ("dw_stencil", _half(lambda a: _dw(a, (1, 1)))),
    "tsii_bf16_avgpool": ("dw_stencil", lambda a: (2.0 * 2 * a[0] * a[1] * a[2] * a[3], 0.0)),
    "tsii_bf16_dense_fwd": ("dense_conv", _half(lambda a: _dense(a, 1, 0, 0, 1))),
    "tsii_bf16_dense_bwd_dx": ("dense_conv", _half(lambda a: _dense(a, 0, 1, 1, 0))),
    "tsii_bf16_dense_bwd_dw": ("dense_conv", _half(lambda a: _dense(a, 1, 1, 0, 0))),
    "tsii_bf16_bn_act_fwd": ("bn_act", _half(lambda a: _bn(a, 2))), "tsii_bf16_bn_stats": ("bn_act", _half(lambda a: _bn(a, 1))),
    "tsii_bf16_bn_act_bwd": ("bn_bwd", _half(lambda a: _bn(a, 3 if a[6] else 5))),      # rows == 0: the kernel takes its own reduction pass (2 more reads)
})
BOUND = {"gemm_nt": None, "gemm_tn": None, "dense_conv": "mfma", "dw_stencil": "hbm", "bn_act": "hbm", "bn_bwd": "hbm", "upcat": "hbm"}


def class_table(timed, steps, products):
    """{class: {ms_per_step, alg GB/step, TB/s, frac of 8 TB/s [, TFLOP/s, frac of the MFMA peak of the arithmetic mode]}}"""
    agg = {}
    for name, recs in timed.items():
        if name not in CALLS:
            continue
        cls, fn = CALLS[name]
        for ms, a in recs:
            try:
                by, macs = fn(a)
            except (IndexError, TypeError):
                continue
            d = agg.setdefault(cls, {"ms": 0.0, "bytes": 0.0, "macs": 0.0, "launches": 0, "floor": 0.0})
            d["ms"] += ms; d["bytes"] += by; d["macs"] += macs; d["launches"] += 1
            # this launch's own roofline: the larger of its HBM time and (matrix kernels) its MFMA time in the arithmetic mode --
            # a class mixes HBM-bound layers (few channels) and MFMA-bound ones, so the class-level fractions under-state both
            mfma_peak = (PEAK_BF16_TFLOPS / products if products else PEAK_FP32_TFLOPS) * 1e12
            d["floor"] += max(by / (PEAK_HBM_TBS * 1e12), (2.0 * macs / mfma_peak) if cls.startswith(("gemm", "dense")) else 0.0) * 1e3
    out = {}
    for cls, d in agg.items():
        if d["ms"] <= 0:
            continue
        t = d["ms"] * 1e-3
        tbs = d["bytes"] / t / 1e12
        rec = {"ms_per_step": round(d["ms"] / steps, 3), "launches_per_step": d["launches"] // steps,
               "alg_gb_per_step": round(d["bytes"] / steps / 1e9, 3), "tb_per_s": round(tbs, 3), "hbm_frac": round(tbs / PEAK_HBM_TBS, 4),
               "per_launch_roofline_ms": round(d["floor"] / steps, 3), "per_launch_roofline_frac": round(d["floor"] / d["ms"], 4)}
        if d["macs"] > 0 and cls.startswith(("gemm", "dense")):
            tf = 2.0 * d["macs"] / t / 1e12
            split = products != 0      # point-wise GEMMs and the implicit-GEMM dense convolutions both run on the split kernels
            # matrix-core work actually issued: `products` bf16 MFMA partial products per fp32 product in the split modes
            peak = PEAK_BF16_TFLOPS / products if split else PEAK_FP32_TFLOPS
            rec.update({"fp32_equiv_tflops": round(tf, 2), "mfma_peak_fp32_equiv": round(peak, 1), "mfma_frac": round(tf / peak, 4)})
        out[cls] = rec
    return out


def pmc_traffic(kernel_prefix, summary=None, all_files=False):
    """HBM bytes per launch of the kernels whose names start with (one of) kernel_prefix, from the committed PMC summary -- or
    (None, why) when there is none for the sources as they are now."""
    summary = summary or PMC_SUMMARY
    path = os.path.join(ROOT, "profiles", summary)
    try:
        js = json.load(open(path))
        if js.get("csrc_sha") != csrc_sha(all_files):
            return None, f"profiles/{summary} was measured at csrc {js.get('csrc_sha')}, sources are now {csrc_sha(all_files)}: stale, not reported"
        prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
        recs = [r for k, r in js["kernels"].items() if k.startswith(prefixes)]
        launches = sum(r["launches"] for r in recs)
        return sum(r["bytes_per_launch"] * r["launches"] for r in recs) / launches, f"profiles/{summary} (csrc {js['csrc_sha']})"
    except Exception as exc:  # noqa: BLE001 - no summary committed for this kernel/config
        return None, f"no PMC summary ({type(exc).__name__})"


def host_cpu():
    model, phys = "unknown", None
    try:
        cores = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys, os.cpu_count()


def cpu_baseline(size: int, threads: int):
    """Oracle (stock-PyTorch CPU restatement of the reference) on a bounded sample of the workload, SURVEY.md 8(d)
    protocol: bs 4, 1 warm-up + 3 timed steps, median."""
    from oracle import pconv_oracle as O
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    import text_segmentation_image_inpainting_amd as T
    model, phys, logical = host_cpu()
    torch.set_num_threads(threads)
    bs = 4
    torch.manual_seed(0)
    ref_like = T.ImageFill()  # parameter container only (default init); the oracle does the math on CPU
    sd = {k: v.detach().clone() for k, v in ref_like.state_dict().items()}
    for k, p in ref_like.named_parameters():
        if p.requires_grad:
            sd[k].requires_grad_(True)
    corrupted, mask, clean = make_batch(bs, size, seed0=10_000)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        out = O.image_fill(sd, corrupted, mask, training=True)
        loss = O.l1_mean(out, clean)
        loss.backward()
        for v in sd.values():
            if v.grad is not None:
                v.grad = None
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[1]
    return {"value": round(bs / t, 4), "unit": "imgs/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
            "sample": f"ImageFill {size}x{size} bs {bs} fwd+bwd (train-mode BN, L1 loss), 1 warm-up + 3 timed steps, median; "
                      f"{threads} torch threads"}


# tests/test_bench_multiprocess.py sets this to {"device": cpu, "backend": "gloo"} (inside the kernel emulator's context) to run the
# N > 1 launch / timing / reporting path of this file on the CPU container; never set in production, where the absence of a ROCm
# GPU is fatal.
TEST_RUNTIME = None


# the secondary BASELINE configs the default line also times (short legs, rank 0, N = 1): name -> argv
SECONDARY_LEGS = {
    "cfg5_xceptiontextsegment_1024_bs8_bf16_storage": ["--model", "XceptionTextSegment", "--size", "1024", "--batch", "8", "--storage", "bf16"],
    "cfg3_textsegament_512_bs64_pixel_shuffle": ["--model", "TextSegament", "--size", "512", "--batch", "64", "--pixel-shuffle"],
}


def main(argv=None, return_line=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--model", default="ImageFill", choices=["ImageFill", "ImageFillOrigin", "ImageFillOriginV2", "TextSegament", "XceptionTextSegment"],
                    help="ImageFill = the headline workload (BASELINE configs[1]); the segmentation nets are the secondary configs "
                         "(cfg 3: --model TextSegament --batch 64 --pixel-shuffle; cfg 5: --model XceptionTextSegment --size 1024 --batch 8 --products 1)")
    ap.add_argument("--pixel-shuffle", action="store_true", help="TextSegament with the Conv(128,16) + PixelShuffle(4) head (cfg 3)")
    ap.add_argument("--checkpoint", action="store_true", help="TextSegament: recompute the encoder stages in backward (memory saver)")
    ap.add_argument("--products", type=int, default=-1, help="tsii_set_gemm_products for this run (1 = the 'mixed bf16' arithmetic of cfg 5)")
    ap.add_argument("--storage", default="f32", choices=["f32", "bf16"],
                    help="activation storage (segmentation nets): bf16 = activations and their gradients in HBM as bf16, fp32 accumulation / statistics / "
                         "parameters (cfg 5: --model XceptionTextSegment --size 1024 --batch 8 --storage bf16)")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B switch of the package (module attribute of ops / partial_convolution, e.g. FUSE_UPCAT=0, FUSE_BN=stats); repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the short run in the bit-exact f32-MFMA arithmetic mode")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU-baseline leg (default: both min(physical cores, 32) and all physical cores, the better one reported)")
    ap.add_argument("--bernoulli-masks", action="store_true", help="stress variant: i.i.d. per-channel masks")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short cfg 5 / cfg 3 legs the default headline run appends (secondary_configs)")
    ap.add_argument("--graph", action="store_true", help="also replay the step from a HIP graph (measured: no gain; off by default)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = TEST_RUNTIME is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = TEST_RUNTIME["device"]
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if on_gpu:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(TEST_RUNTIME["backend"])
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer

    for kv in args.knob:
        from text_segmentation_image_inpainting_amd import ops as _ops, partial_convolution as _pc
        name, _, val = kv.partition("=")
        mod = _ops if hasattr(_ops, name) else _pc if hasattr(_pc, name) else None
        if mod is None:
            raise SystemExit(f"--knob {name}: no such switch")
        cur = getattr(mod, name)
        setattr(mod, name, (val not in ("0", "false", "False")) if isinstance(cur, bool) else val)
    if on_gpu:
        _lib.lib()           # fails loudly when libtsii_hip.so is missing
    if args.products >= 0:
        _lib.set_gemm_products(args.products)
    products = _lib.get_gemm_products()
    bf16_storage = args.storage == "bf16"
    if bf16_storage:
        if args.model != "XceptionTextSegment":
            raise SystemExit("--storage bf16 exists for XceptionTextSegment (BASELINE config 5): the partial-convolution family keeps fp32 storage, and "
                             "TextSegament's scSE / global-average-pool / pixel-shuffle layers have no bf16 kernels")
        T.set_activation_storage("bf16")
        products = 1          # accounting only: the tsii_bf16_* matrix products are one bf16 MFMA product per multiply-add
        args.no_f32_leg = True
    torch.manual_seed(0)  # identical random-init weights on every rank
    seg = args.model in ("TextSegament", "XceptionTextSegment")
    if seg:
        # secondary workloads: logits = net(image), BinaryFocalLoss against the text mask (train.py of the reference); the
        # trainer sees the same (inputs, mask, target) step signature through a one-line adapter
        from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
        # (the CPU test of the N > 1 path swaps in a three-layer network through the same TEST_RUNTIME["model_factory"] as below)
        net = ((TEST_RUNTIME or {}).get("model_factory") or
               (lambda: getattr(T, args.model)(**({"pixel_shuffle_head": True} if (args.pixel_shuffle and args.model == "TextSegament") else {}))))()
        if args.checkpoint and hasattr(net, "checkpoint_encoder"):
            net.checkpoint_encoder = True

        class SegStep(torch.nn.Module):
            def __init__(self, net):
                super().__init__()
                self.net = net

            def forward(self, args_):
                return self.net(args_[0])
        model = SegStep(net).to(dev).train()
        focal = T.BinaryFocalLoss(0, 1, 2)
        trainer = FlatSGDTrainer(model, lr=1e-3, momentum=0.9, weight_decay=1e-4, loss_fn=lambda out, tgt: focal(out, tgt))
        trainer.broadcast_parameters()
        corrupted, clean_nhwc = (t.to(dev) for t in make_seg_batch(args.batch, args.size, seed0=rank * args.batch))
        mask = None
    else:
        # (the CPU test of the N > 1 path swaps in a two-block network: a whole ImageFill step takes minutes on the kernel emulator)
        factory = (TEST_RUNTIME or {}).get("model_factory") or (lambda: getattr(T, args.model)())
        model = factory().to(dev).train()
        trainer = FlatSGDTrainer(model, lr=1e-3, momentum=0.9, weight_decay=1e-4, **((TEST_RUNTIME or {}).get("trainer_kwargs") or {}))
        trainer.broadcast_parameters()
        corrupted, mask, clean = make_batch(args.batch, args.size, seed0=rank * args.batch, bernoulli=args.bernoulli_masks)
        corrupted, mask = corrupted.to(dev), mask.to(dev)   # inputs resident in HBM before the timed region
        clean_nhwc = to_nhwc(clean.to(dev))

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
    # HIP events around entry points need the GPU's streams; the CPU run of the N > 1 path reports no kernel classes
    start_timing = _lib.start_timing if on_gpu else (lambda names: None)
    stop_timing = _lib.stop_timing if on_gpu else (lambda: {})

    gemm_calls = [n for n, (c, _) in CALLS.items() if c.startswith("gemm")]
    # the storage type the network ACTUALLY runs in: which family of entry points its first step calls (what the line reports)
    _lib.count_calls(True)
    loss = trainer.step(corrupted, mask, clean_nhwc)
    called = _lib.count_calls(False)
    seen_dtypes = {"bf16" if n.startswith("tsii_bf16_") else "f32" for n in called if n.startswith(("tsii_bf16_pw", "tsii_bf16_dw", "tsii_bf16_dense", "tsii_pw", "tsii_dw", "tsii_dense"))}
    storage_seen = "bf16" if "bf16" in seen_dtypes else "f32"
    if storage_seen != args.storage:
        raise SystemExit(f"--storage {args.storage} was requested but the network's layers ran the {sorted(seen_dtypes)} convolution kernels: not reporting a line "
                         "whose dtype the run did not have")
    for _ in range(args.warmup - 1):
        loss = trainer.step(corrupted, mask, clean_nhwc)
    sync()
    start_timing(gemm_calls)      # HIP events (launch stream) around the GEMM entry points inside the timed region
    trainer.measure_exposed = world > 1    # N > 1: how long each step stalls for all-reduces backward did not hide (2 events / step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(corrupted, mask, clean_nhwc)
    sync()
    elapsed = time.perf_counter() - t0
    timed = stop_timing()
    eager_ms = elapsed / args.steps * 1e3
    trainer.measure_exposed = False
    # "comm": bucket layout, stand-alone all-reduce time / bus bandwidth of the whole gradient buffer, and (N > 1) the exposed
    # vs overlapped split of it measured in the timed steps above
    comm = trainer.comm_stats() if hasattr(trainer, "comm_stats") else None
    graphed = False
    if args.graph:
        try:
            trainer.capture(corrupted, mask, clean_nhwc)
            for _ in range(args.warmup):
                loss = trainer.step_graph()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = trainer.step_graph()
            sync()
            elapsed = time.perf_counter() - t0
            graphed = True
        except Exception as exc:  # noqa: BLE001 - capture unsupported here: the eager measurement stands
            print(f"[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); reporting the eager loop", file=sys.stderr)
            sync()
    final_loss = float(loss.item())
    # per-class pass: HIP events around EVERY hot entry point for a few extra steps (outside the timed region: the
    # ~1800 extra event records per step would perturb `value`)
    prof_steps = 3
    start_timing(list(CALLS))
    for _ in range(prof_steps):
        trainer.step(corrupted, mask, clean_nhwc)
    classes = class_table(stop_timing(), prof_steps, products)
    # forward-only rate (SURVEY.md 8(d) asks for both): train-mode BatchNorm forward + loss, no autograd tape
    fwd_steps = max(2, args.steps // 2)
    with torch.no_grad():
        trainer.loss_fn(model((corrupted, mask)), clean_nhwc)
        sync()
        t1 = time.perf_counter()
        for _ in range(fwd_steps):
            trainer.loss_fn(model((corrupted, mask)), clean_nhwc)
        sync()
        fwd_elapsed = time.perf_counter() - t1
    # the same step in the other arithmetic modes, short runs, for reference: 0 = bit-exact f32-MFMA (fp32 FMA chain),
    # 3 = 2-piece split / 3 partial products (opt-in: ~8x the rounding error of the fp32 chain, still 3 orders inside the 1e-3 bar)
    def mode_leg(mode, what):
        _lib.set_gemm_products(mode)
        for _ in range(3):
            trainer.step(corrupted, mask, clean_nhwc)
        sync()
        t2 = time.perf_counter()
        n2 = max(5, args.steps // 5)
        for _ in range(n2):
            trainer.step(corrupted, mask, clean_nhwc)
        sync()
        e2 = time.perf_counter() - t2
        with torch.no_grad():
            t3 = time.perf_counter()
            for _ in range(n2):
                trainer.loss_fn(model((corrupted, mask)), clean_nhwc)
            sync()
            e3 = time.perf_counter() - t3
        _lib.set_gemm_products(products)
        return {"value": round(world * args.batch * n2 / e2, 2), "unit": "imgs/s (rank 0 clock)", "ms_per_step": round(e2 / n2 * 1e3, 3), "steps": n2,
                "forward_ms_per_step": round(e3 / n2 * 1e3, 3), "arithmetic": what}
    f32_leg = split3_leg = None
    if products != 0 and not args.no_f32_leg:
        f32_leg = mode_leg(0, "v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chain)")
        if products != 3:
            split3_leg = mode_leg(3, "2-piece bf16 split, 3 partial products (dropped terms <= 2^-15 |a*b|; opt-in, NOT the reported arithmetic)")
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        imgs = world * args.batch * args.steps
        value = imgs / elapsed
        # roofline of the dominant kernel class, from the HIP events of the timed region
        tt_classes = class_table(timed, args.steps, products)
        dom = max(tt_classes.items(), key=lambda kv: kv[1]["ms_per_step"]) if tt_classes else None
        roofline = None
        if dom:
            k, d = dom
            # the kernels behind the class: the persistent producer/consumer kernel takes every 6-product layer with N >= 128 that
            # tiles evenly (csrc/gemm_pc.hip nt_pc_ok), the block-synchronous split kernel the rest
            kern = {"gemm_nt": (("tsii::gemm_nt_pc_kernel<", "tsii::gemm_nt_split_kernel<") if products else "tsii::gemm_nt_kernel<2, 2, 2, 2"),
                    "gemm_tn": ("tsii::gemm_tn_split_kernel<2, 2, 2, 2" if products else "tsii::gemm_tn_kernel<2, 2, 2, 2")}.get(k, "tsii::" + k)
            if bf16_storage:
                kern = {"gemm_nt": "tsii::hgemm_nt_kernel<", "gemm_tn": "tsii::hgemm_tn_kernel<", "dw_stencil": "tsii::hdw_conv_kernel< + hdw_dw_kernel<",
                        "dense_conv": "tsii::hgemm_nt_kernel< (gathered) + hgemm_tn_kernel<", "bn_bwd": "tsii::hbn_bwd_apply_kernel + partial",
                        "bn_act": "tsii::hbn_apply_kernel"}.get(k, "tsii::" + k)
            kname = kern if isinstance(kern, str) else " + ".join(x.split("::")[1] + "...>" for x in kern)
            kname = kname.split("::")[1] + "...>" if "::" in kname else kname
            t_hbm = d["alg_gb_per_step"] / (PEAK_HBM_TBS * 1e3)            # seconds per step at the HBM peak
            t_mfma = d["ms_per_step"] * 1e-3 * d.get("mfma_frac", 0.0)      # seconds per step at the MFMA peak of the mode
            hbm_bound = t_hbm >= t_mfma
            if args.batch == 32 and args.size == 512 and args.model == "ImageFill":
                traffic, traffic_src = pmc_traffic(kern)
            elif bf16_storage and args.model == "XceptionTextSegment" and args.size == 1024 and args.batch == 8 and k == "gemm_nt":
                # the 1x1 products only: AMODE (5th / 1st template argument) 0 -- the gathered forms of the same kernels are class dense_conv
                traffic, traffic_src = pmc_traffic(("tsii::hgemm_nt_kernel<2, 2, 2, 4, 0,", "tsii::hgemm_nt_kernel<2, 2, 2, 2, 0,", "tsii::hgemm_nt_kernel<2, 2, 2, 1, 0,",
                                                    "tsii::hgemm_nt_kernel<4, 1, 1, 1, 0,", "tsii::hgemm_nt_ph_kernel<0,"), PMC_SUMMARY_CFG5, all_files=True)
            else:
                traffic, traffic_src = None, "not a profiled configuration"
            roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname + " (class " + k + (": 1x1-conv forward + dX GEMMs)" if k == "gemm_nt" else ")"),
                        "achieved": d["tb_per_s"] * 1e3 if hbm_bound else d["fp32_equiv_tflops"],
                        "peak": PEAK_HBM_TBS * 1e3 if hbm_bound else d["mfma_peak_fp32_equiv"],
                        "unit": "GB/s" if hbm_bound else "TFLOP/s (fp32-equivalent)",
                        "frac": d["hbm_frac"] if hbm_bound else d["mfma_frac"],
                        "hbm_frac": d["hbm_frac"], "mfma_frac": d.get("mfma_frac"),
                        # sum over the class's launches of max(HBM time, MFMA time) / measured: each layer against its own bound
                        "per_launch_roofline_frac": d["per_launch_roofline_frac"],
                        "traffic": traffic, "traffic_source": traffic_src,
                        "alg_bytes_per_launch": round(d["alg_gb_per_step"] * 1e9 / max(1, d["launches_per_step"])),
                        "launches_per_step": d["launches_per_step"],
                        "avg_launch_ms": round(d["ms_per_step"] / max(1, d["launches_per_step"]), 4),
                        "ms_per_step_in_class": d["ms_per_step"],
                        "arithmetic": ("bf16 operands from HBM, one v_mfma_f32_32x32x16_bf16 product per multiply-add, fp32 accumulate, one RNE rounding per stored value" if bf16_storage
                                       else f"split-bf16: {products} v_mfma_f32_32x32x16_bf16 partial products per fp32 product, fp32 accumulate" if products
                                       else "v_mfma_f32_32x32x2_f32")}
        ms_per_img = elapsed / imgs * world * 1e3  # per-GPU ms per image
        whole = None
        if (args.model, args.size) in WHOLE_FWD:
            gf, gb = WHOLE_FWD[(args.model, args.size)]
            gb = gb / 2 if bf16_storage else gb           # bf16 activation storage: the same tensors at 2 bytes per element
            flop_peak = PEAK_BF16_TFLOPS if (bf16_storage or products == 1) else PEAK_FP32_TFLOPS     # the arithmetic the mode asks for
            t_hbm = 3 * gb / (PEAK_HBM_TBS * 1e3) * 1e3
            t_flop = 3 * gf / (flop_peak * 1e3) * 1e3
            fwd_ms_per_img = fwd_elapsed / fwd_steps / args.batch * 1e3
            whole = {"alg_gflop_per_img": round(3 * gf, 1), "alg_gb_per_img": round(3 * gb, 3),
                     "t_min_ms_per_img_hbm": round(t_hbm, 3),
                     ("t_min_ms_per_img_bf16_flop" if flop_peak == PEAK_BF16_TFLOPS else "t_min_ms_per_img_fp32_flop"): round(t_flop, 3),
                     "frac_of_roofline": round(max(t_hbm, t_flop) / ms_per_img, 4),
                     "frac_of_hbm_roofline": round(t_hbm / ms_per_img, 4),
                     "forward_frac_of_roofline": round(max(t_hbm, t_flop) / 3 / fwd_ms_per_img, 4),
                     "forward_frac_of_hbm_roofline": round(t_hbm / 3 / fwd_ms_per_img, 4)}
        if seg:
            workload = (f"{args.model}{' (pixel-shuffle head)' if args.pixel_shuffle and args.model == 'TextSegament' else ''}"
                        f"{' (encoder stages recomputed in backward)' if args.checkpoint else ''} {args.size}x{args.size} text-segmentation train step "
                        f"(fwd+bwd, train-mode BN, BinaryFocalLoss, grad all-reduce, fused SGD), {args.batch} imgs/GPU, synthetic manga tiles")
            metric = f"imgs/sec fwd+bwd on {args.size}x{args.size} text segmentation ({args.model}; secondary config, not the headline metric)"
        else:
            workload = (f"{args.model} {args.size}x{args.size} partial-conv inpainting train step (fwd+bwd, train-mode BN, L1 loss, grad all-reduce, "
                        f"fused SGD), {args.batch} imgs/GPU, random line/ellipse hole masks")
            metric = "imgs/sec fwd+bwd on 512x512 partial-conv inpaint"
        dtype = "bf16 activation / activation-gradient storage (NHWC, 8 channels per 16-byte vector); fp32 accumulation, BatchNorm statistics, parameters and parameter gradients; matrix products bf16 x bf16 -> fp32 (BASELINE config 5 'mixed bf16')" if bf16_storage else {0: "f32", 1: "f32 storage / accumulation / stencils / BatchNorm; matrix products on bf16-rounded operands (the 'mixed bf16' arithmetic of cfg 5)"}.get(
            products, f"f32 (storage, accumulation, stencils, BatchNorm; matrix products = {products}-term exact bf16 split on the bf16 MFMA, fp32-class error)")
        line = {
            "metric": metric,
            "value": round(value, 2), "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * args.batch, "parallelism": f"dp{world}", "gemm_products": products,
                       "activation_storage": storage_seen},
            "roofline": roofline, "kernel_classes": classes, "whole_step_roofline": whole, "final_loss": final_loss,
            "launch": "hip_graph_replay" if graphed else "eager", "eager_ms_per_step": round(eager_ms, 3),
            "forward_only": {"value": round(world * args.batch * fwd_steps / fwd_elapsed, 2), "unit": "imgs/s (rank 0 clock)",
                             "ms_per_step": round(fwd_elapsed / fwd_steps * 1e3, 3), "steps": fwd_steps},
            "f32_mfma_mode": f32_leg, "split3_mode": split3_leg, "comm": comm,
        }
        line["peak_mem_gib"] = round(torch.cuda.max_memory_allocated() / 2**30, 1) if on_gpu else None
        if not args.no_cpu_baseline and world == 1 and not seg:
            _, phys, logical = host_cpu()
            if args.cpu_threads:
                line["cpu_baseline"] = cpu_baseline(args.size, args.cpu_threads)
            else:
                # two thread counts: 32 (where a 4-image batch stops scaling) and every physical core; the better one is the baseline
                runs = [cpu_baseline(args.size, t) for t in sorted({min(phys or logical or 1, 32), phys or logical or 1})]
                best = max(runs, key=lambda r: r["value"])
                best["all_runs"] = [{"cores": r["cores"], "value": r["value"]} for r in runs]
                line["cpu_baseline"] = best
        elif world == 1:
            line["cpu_baseline"] = None
        if return_line:
            return line
        headline = (args.model == "ImageFill" and args.size == 512 and args.batch == 32 and args.products < 0 and not bf16_storage
                    and not args.knob and not args.bernoulli_masks)
        if headline and world == 1 and on_gpu and not args.no_secondary:
            # BASELINE configs 5 and 3 through the same code path, short legs (like f32_mfma_mode) so that every default run -- the
            # driver's included -- times them; the headline's tensors are released first
            import gc
            del model, trainer, corrupted, mask, clean_nhwc, loss
            gc.collect()
            torch.cuda.empty_cache()
            line["secondary_configs"] = {name: secondary_leg(argv2) for name, argv2 in SECONDARY_LEGS.items()}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def secondary_leg(argv2, steps=6, warmup=3):
    """One secondary config as a short run of main() in this process: the fields of its line that say how fast and against which bound."""
    import gc
    from text_segmentation_image_inpainting_amd import _lib
    import text_segmentation_image_inpainting_amd as T
    torch.cuda.reset_peak_memory_stats()
    try:
        rec = main(list(argv2) + ["--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-f32-leg", "--no-secondary"], return_line=True)
        out = {"argv": " ".join(argv2), "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": steps, "warmup": warmup,
               "forward_ms_per_step": rec["forward_only"]["ms_per_step"], "dtype": rec["dtype"], "workload": rec["config"]["workload"],
               "whole_step_roofline": rec["whole_step_roofline"], "roofline": rec["roofline"], "peak_mem_gib": rec["peak_mem_gib"],
               "kernel_classes": {k: {f: v[f] for f in ("ms_per_step", "hbm_frac", "mfma_frac", "per_launch_roofline_frac") if f in v}
                                  for k, v in (rec["kernel_classes"] or {}).items()}}
    except Exception as exc:  # noqa: BLE001 - a secondary leg must never take the headline line down with it
        out = {"argv": " ".join(argv2), "error": f"{type(exc).__name__}: {exc}"}
    finally:
        T.set_activation_storage("f32")
        _lib.set_gemm_products(None)
        gc.collect()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()

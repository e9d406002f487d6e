"""CPU-side checks: the C-ABI library loads and exports exactly what include/tsii_hip.h declares,
the ctypes table mirrors the header, the module mirror keeps the reference's state_dict layout,
and the product path refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import text_segmentation_image_inpainting_amd as T
from text_segmentation_image_inpainting_amd import _lib, build_ext, masks, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "tsii_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsii_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_ctypes_table():
    assert _header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    lib = build_ext.build(verbose=False)  # hipcc cross-compiles for gfx950 without a GPU
    cdll = ctypes.CDLL(lib)
    for name in _header_symbols():
        assert hasattr(cdll, name), f"{name} declared in include/tsii_hip.h but not exported"
    _lib.bind(cdll)
    assert cdll.tsii_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "tsii_hip.h")).read()
    assert int(re.search(r"#define\s+TSII_ABI_VERSION\s+(\d+)", header).group(1)) == _lib.ABI_VERSION


def test_state_dict_layout_matches_reference(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for name in ("ImageFill", "ImageFillOrigin", "ImageFillOriginV2"):
        m = getattr(T, name)()
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == keys[name]
        assert [k for k, p in m.named_parameters() if p.requires_grad] == keys[name + ".trainable"]
        for k, v in m.state_dict().items():
            if k.endswith("mask_conv.weight"):
                assert bool((v == 1).all())


def test_no_cpu_fallback():
    m = T.PartialConv(3, 4, 3, 1, 1)
    x = torch.randn(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path|GPU"):
        m((x, torch.ones_like(x)))


def test_tolerant_load_state_dict(capsys):
    m = T.PartialConv(3, 4, 3, 1, 1)
    sd = {"feature_conv.weight": torch.zeros(4, 3, 3, 3), "nonexistent.key": torch.zeros(1),
          "feature_conv.bias": torch.zeros(5)}
    unknown, failed = m.load_state_dict(sd)  # never raises (models/BaseModels.py:41-52): reports and continues
    out = capsys.readouterr().out
    assert unknown == ["nonexistent.key"] and failed == ["feature_conv.bias"]
    assert "nonexistent.key" in out and "feature_conv.bias" in out
    assert float(m.feature_conv.weight.detach().abs().sum()) == 0.0


def test_mask_parts_roundtrip():
    plane = (torch.rand(2, 5, 6) > 0.5).float()
    t = plane.unsqueeze(1).expand(-1, 7, -1, -1)
    mp = masks.MaskParts.from_tensor(t)
    assert mp.planar and mp.channels == 7
    back = mp.as_tensor()
    assert back.shape == t.shape and back.stride(1) == 0 and torch.equal(back, t)
    full = (torch.rand(2, 3, 5, 6) > 0.5).float()
    mp2 = masks.MaskParts.from_tensor(full)
    assert not mp2.planar and torch.equal(mp2.as_tensor(), full)


def test_synthetic_dataset_contract():
    c, m, cl = synthetic.make_batch(2, 128, seed0=3)
    assert c.shape == m.shape == cl.shape == (2, 3, 128, 128) and c.dtype == torch.float32
    assert set(np.unique(m.numpy())) <= {0.0, 1.0}
    assert torch.equal(c, cl * m)                                # Dataloader.py:131
    assert torch.equal(m[:, 0], m[:, 1]) and torch.equal(m[:, 0], m[:, 2])
    c2, m2, _ = synthetic.make_batch(2, 128, seed0=3)
    assert torch.equal(m, m2) and torch.equal(c, c2)             # seeded


def test_dilate_known_answer():
    """10x10 dilation, anchor (5,5): a single pixel grows to rows/cols -4..+5 around it."""
    a = np.zeros((32, 32), np.uint8)
    a[16, 16] = 255
    d = synthetic.dilate_10x10(a)
    ys, xs = np.nonzero(d)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (12, 21, 12, 21) and d.sum() == 255 * 100


def test_bench_class_table_and_traffic_gate(tmp_path, monkeypatch):
    """bench.py host logic: per-class accounting from the timed entry-point records (algorithmic bytes / MACs decoded from
    the call arguments, fractions of the 8 TB/s and of the MFMA peak of the arithmetic mode) and the staleness gate of the
    PMC traffic figure (reported only while the kernel sources hash to what the profile was measured at)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    m, k, n = 1 << 20, 256, 512
    timed = {"tsii_pw_fwd": [(1.0, (m, k, n))] * 2,                       # 2 launches of 1 ms in 2 steps
             "tsii_bn_act_fwd": [(0.5, (m, n))] * 2, "not_a_kernel": [(9.0, ())]}
    tab = bench.class_table(timed, 2, 6)
    g = tab["gemm_nt"]
    assert g["launches_per_step"] == 1 and g["ms_per_step"] == 1.0
    assert abs(g["alg_gb_per_step"] - 4.0 * m * (k + n) / 1e9) < 1e-2
    assert abs(g["fp32_equiv_tflops"] - 2.0 * m * k * n / 1e-3 / 1e12) < 0.5
    assert abs(g["mfma_peak_fp32_equiv"] - bench.PEAK_BF16_TFLOPS / 6) < 0.1 and 0 < g["mfma_frac"] < 1
    assert abs(bench.class_table(timed, 2, 0)["gemm_nt"]["mfma_peak_fp32_equiv"] - bench.PEAK_FP32_TFLOPS) < 0.1
    b = tab["bn_act"]
    assert abs(b["tb_per_s"] - 8.0 * m * n / 0.5e-3 / 1e12) < 0.05 and "mfma_frac" not in b
    # per-launch roofline: every launch against the larger of ITS HBM time and (matrix kernels) ITS MFMA time
    t_hbm = 4.0 * m * (k + n) / (bench.PEAK_HBM_TBS * 1e12) * 1e3
    t_mfma = 2.0 * m * k * n / (bench.PEAK_BF16_TFLOPS / 6 * 1e12) * 1e3
    assert abs(g["per_launch_roofline_ms"] - max(t_hbm, t_mfma)) < 2e-3 and abs(g["per_launch_roofline_frac"] - max(t_hbm, t_mfma) / 1.0) < 2e-3
    assert abs(b["per_launch_roofline_frac"] - b["hbm_frac"]) < 1e-3              # a streaming class: the two coincide
    # a class mixing an HBM-bound and an MFMA-bound layer: the blended fraction exceeds both class-level fractions
    mix = bench.class_table({"tsii_pw_fwd": [(1.0, (m, 1024, 1024)), (1.0, (8 * m, 64, 32))]}, 1, 6)["gemm_nt"]
    assert mix["per_launch_roofline_frac"] > max(mix["hbm_frac"], mix["mfma_frac"])
    # the K4c entry points are accounted for (virtual concatenation: bytes of low + skip + y only)
    hc = bench.class_table({"tsii_head_cat_fwd": [(0.5, (32, 3, 32, 512, 512, 3, 5184))]}, 1, 6)["dense_conv"]
    assert abs(hc["alg_gb_per_step"] - 4.0 * 32 * (512 * 512 // 4 * 32 + 512 * 512 * 3 + 512 * 512 * 3) / 1e9) < 1e-2
    # ... and so are the matrix-core head entries (K4d: same bytes) and the BatchNorm backward that also pools the K7b addend gradient
    for name in ("tsii_head_cat_fwd_low", "tsii_head_cat_bwd_dw_low", "tsii_head_cat_bwd_low"):
        hl = bench.class_table({name: [(0.2, (32, 3, 32, 512, 512, 3, 5184))]}, 1, 6)["dense_conv"]
        assert abs(hl["alg_gb_per_step"] - hc["alg_gb_per_step"]) < 1e-6, name
    bp = bench.class_table({"tsii_bn_act_bwd_pre_pool": [(1.0, (m, n, 1e-5, 2, 0.3, 1, 512, 256, 256, 4096))]}, 1, 6)["bn_bwd"]
    assert abs(bp["alg_gb_per_step"] - 4.0 * m * n * 3.25 / 1e9) < 1e-2
    # several kernels behind one class in the traffic lookup
    # traffic gate
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    js = {"csrc_sha": "0" * 12, "kernels": {"tsii::gemm_nt_split_kernel<2, 2, 2, 2, 6>": {"launches": 2, "bytes_per_launch": 1e9}}}
    json.dump(js, open(tmp_path / "profiles" / bench.PMC_SUMMARY, "w"))
    monkeypatch.setattr(bench, "csrc_sha", lambda *a: "1" * 12)
    val, why = bench.pmc_traffic("tsii::gemm_nt_split_kernel<2, 2, 2, 2")
    assert val is None and "stale" in why
    monkeypatch.setattr(bench, "csrc_sha", lambda *a: "0" * 12)
    val, why = bench.pmc_traffic("tsii::gemm_nt_split_kernel<2, 2, 2, 2")
    assert val == 1e9
    js["kernels"]["tsii::gemm_nt_pc_kernel<1, 8, 6, false, 0, 0>"] = {"launches": 6, "bytes_per_launch": 2e9}
    json.dump(js, open(tmp_path / "profiles" / bench.PMC_SUMMARY, "w"))
    val, why = bench.pmc_traffic(("tsii::gemm_nt_pc_kernel<", "tsii::gemm_nt_split_kernel<"))
    assert abs(val - (2 * 1e9 + 6 * 2e9) / 8) < 1.0


# Kernels that are allowed private scratch (register spills), by mangled-name prefix -> largest size in bytes per lane.  Every entry
# was measured: the fused lean depth-wise strips and the head's weight-gradient forms at 2 waves per SIMD WITHOUT scratch run no
# faster than at 3 waves with it (profiles/r05i_spill_ab.log: ImageFill 480.8 vs 480.1 img/s, ImageFillOrigin 395.1 vs 394.8, the
# strips' own timings within 1 %) -- the spilled values live in the epilogues, not in the steady-state loops; the GEMM entries are
# 8-16 bytes in their epilogues.  Anything else -- in particular every bf16-storage kernel and the headline instantiations of the
# head (NB = 2, low-tensor mask, d low in the same pass) -- must compile without scratch.
SCRATCH_ALLOWED = {
    "_ZN4tsii14dw_lean_kernelILi1ELb0E": 48, "_ZN4tsii14dw_lean_kernelILi2ELb1E": 40,
    # K6e with mask planes (round 6): 2-4 registers at the 256-register limit of 2 waves per SIMD; the form was measured WITH them
    # (profiles/r06r_bench_k6e_ab.log: step 58.4-58.6 ms with the kernel, 60.4 without)
    "_ZN4tsii14dw_lean_kernelILi4ELb1ELb1E": 24,
    # K6d on the dilated ring kernels (dilation 2 / 4; ImageFill's 32 x 32 levels, 0.2 ms per step each): 14-18 registers at the
    # 256-register limit; the form was measured WITH them (profiles/r06v_bench_dil_ab.log: 56.9-57.1 vs 57.45-57.48 ms without the forms)
    "_ZN4tsii15dw_strip_kernelILi1ELi2ELi3E": 80, "_ZN4tsii15dw_strip_kernelILi1ELi4ELi3E": 64,
    "_ZN4tsii17gemm_nt_pc_kernelILi2ELi4ELi6ELb1E": 8,
    "_ZN4tsii20gemm_nt_split_kernelILi2ELi2ELi2ELi2E": 16, "_ZN4tsii20gemm_tn_split_kernelILi2ELi2ELi2ELi2E": 16,
    "_ZN4tsii23head_cat_dw_mfma_kernelILi2ELb0ELb1E": 96, "_ZN4tsii23head_cat_dw_mfma_kernelILi2ELb1ELb0ELb0E": 52,
    "_ZN4tsii23head_cat_dw_mfma_kernelILi2ELb1ELb1E": 116, "_ZN4tsii23head_cat_dw_mfma_kernelILi4E": 152,
}


def test_no_kernel_spills_outside_the_measured_list():
    """Build-time resource records (build_ext.kernel_resources, from -Rpass-analysis=kernel-resource-usage of the very compile that
    produced the library): scratch only where it was measured to be harmless; none in the bf16-storage kernels."""
    from text_segmentation_image_inpainting_amd import build_ext
    res = build_ext.kernel_resources()
    assert sum(len(v) for v in res.values()) > 300
    offenders = []
    for src, kernels in res.items():
        for name, rec in kernels.items():
            sc = rec.get("scratch", 0)
            if sc <= 0:
                continue
            allowed = max((v for k, v in SCRATCH_ALLOWED.items() if name.startswith(k)), default=0)
            if src.startswith("bf16_") or sc > allowed:
                offenders.append((src, name[:90], sc, allowed))
    assert not offenders, offenders
    clean = "_ZN4tsii23head_cat_dw_mfma_kernelILi2ELb1ELb0ELb1E"      # the headline head: weight gradient + d low
    assert any(n.startswith(clean) and r.get("scratch", 1) == 0 for n, r in res["head_mfma.hip"].items())

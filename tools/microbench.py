#!/usr/bin/env python3
"""The two machine numbers SURVEY.md 8(d) asks to be measured on the box, through the library's own kernels:
  * stream: tsii_act_fwd (1 read + 1 write) and tsii_bn_stats (read only) over a 3.2 GB tensor -> GB/s
  * MFMA:   tsii_pw_fwd on 65536 x 4096 x 4096 and 8192^3-like shapes in the f32-MFMA, 6-product split and plain-bf16 arithmetic -> TFLOP/s
    python tools/microbench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    m, c = 2097152, 384
    x = torch.randn(m, c, device=dev)
    y = torch.empty_like(x)
    gb = x.numel() * 4 / 1e9
    ms = timeit(lambda: call("tsii_act_fwd", ptr(x), x.numel(), 2, 0.3, ptr(y), st))
    print(f"stream copy (act_fwd, {gb:.1f} GB read + {gb:.1f} GB write): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
    mean, var = torch.empty(c, device=dev), torch.empty(c, device=dev)
    nb = L.tsii_bn_ws_bytes(m, c)
    ws = torch.empty(nb // 4 + 4, device=dev)
    ms = timeit(lambda: call("tsii_bn_stats", ptr(x), m, c, ptr(mean), ptr(var), None, None, 0.1, ptr(ws), nb, st))
    print(f"stream read (bn_stats, {gb:.1f} GB): {ms:.3f} ms  {gb / ms:.2f} TB/s")
    for (M, K, N) in ((65536, 4096, 4096), (16384, 8192, 8192), (262144, 1024, 1024)):
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.02
        o = torch.empty(M, N, device=dev)
        wws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
        for mode, label, peak in ((0, "f32 MFMA (v_mfma_f32_32x32x2_f32)", 157.3), (6, "split-bf16, 6 products (v_mfma_f32_32x32x16_bf16)", 2500.0 / 6),
                                  (1, "bf16 operands, 1 product", 2500.0)):
            _lib.set_gemm_products(mode)
            ms = timeit(lambda: call("tsii_pw_fwd", ptr(a), M, K, ptr(w), N, None, None, 0, None, None, None, ptr(o), ptr(wws), wws.numel() * 4, st), iters=3)
            tf = 2.0 * M * K * N / ms / 1e9
            print(f"GEMM {M} x {K} x {N} {label}: {ms:.3f} ms  {tf:.1f} TFLOP/s fp32-equivalent ({tf / peak * 100:.0f} % of {peak:.1f})")
        _lib.set_gemm_products(None)

if __name__ == "__main__":
    main()

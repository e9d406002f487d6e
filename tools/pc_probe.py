#!/usr/bin/env python3
"""One GEMM shape through tsii_pw_fwd, a few launches: the thing rocprofv3 counter passes wrap.
    python tools/pc_probe.py M K N [iters]
The TSII_GEMM_PC* knobs it prints only exist in an A/B build of the library:
    python tools/variants/build_variant.py abl gemm_pc.hip -DTSII_GEMM_PC_ABLATIONS
    TSII_LIBRARY=tools/variants/_bin/libtsii_abl.so TSII_GEMM_PC_ABL=128 python tools/pc_probe.py 65536 1024 1024"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    y = torch.empty(M, N, device=dev)
    wws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
    fn = lambda: call("tsii_pw_fwd", ptr(x), M, K, ptr(w), N, None, None, 0, None, None, None, ptr(y), ptr(wws), wws.numel() * 4, st)
    for _ in range(20):      # clocks up before the timed launches
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters
    if int(os.environ.get("TSII_GEMM_PC_ABL", "0")) & 32768:
        d = y.view(-1)[:4096].view(-1, 8, 2).cpu()
        frac = (d[..., 0] / d[..., 1])
        print("consumer waves: cycles waiting for full flags / total: mean %.3f min %.3f max %.3f; total cycles mean %.0f" % (frac.mean(), frac.min(), frac.max(), d[..., 1].mean()))
    if int(os.environ.get("TSII_GEMM_PC_ABL", "0")) & 32768:
        d = y.view(-1)[8192:8192 + 8192].view(-1, 8).cpu()
        print("producer waves: cycles per stage: waiting for an empty slot %.0f | load wait %.0f | split + LDS stores %.0f | load issue %.0f | total %.0f" %
              ((d[:, 0] / d[:, 3]).mean(), (d[:, 4] / d[:, 3]).mean(), (d[:, 1] / d[:, 3]).mean(), (d[:, 5] / d[:, 3]).mean(), (d[:, 2] / d[:, 3]).mean()))
    print(f"M={M} K={K} N={N} opt={os.environ.get('TSII_GEMM_PC_OPT', '')} pc={os.environ.get('TSII_GEMM_PC', '')}: {t:.3f} ms {2.0 * M * K * N / t / 1e9:.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()

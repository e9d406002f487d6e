#!/usr/bin/env python3
"""Micro-benchmark of the point-wise GEMM entry points on ImageFill's layer shapes (bs 32, 512^2).
    python tools/gemm_bench.py [--iters 5] [--only fwd|dx|dw]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

# (M, K, N, masked)  forward shapes; dx / dw are derived
SHAPES = [
    (2097152, 64, 256, False), (2097152, 192, 384, True), (2097152, 384, 32, True),
    (524288, 256, 128, False), (524288, 128, 512, False), (524288, 512, 128, False),
    (524288, 384, 768, True), (524288, 768, 128, True),
    (131072, 256, 1024, False), (131072, 1024, 256, False), (131072, 512, 1024, True),
    (32768, 256, 1024, True), (32768, 1024, 256, True),
    (65536, 1024, 1024, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--modes", default="", help="comma list of tsii_set_gemm_products values (0, 3, 6) to loop over")
    args = ap.parse_args()
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    shapes = SHAPES
    if args.shapes:
        idx = [int(v) for v in args.shapes.split(",")]
        shapes = [SHAPES[i] for i in idx]

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    modes = [int(v) for v in args.modes.split(",")] if args.modes else [L.tsii_get_gemm_products()]
    for (M, K, N, masked), gm in [(sh, m) for sh in shapes for m in modes]:
        _lib.set_gemm_products(gm)
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        dy = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev)
        dx = torch.empty(M, K, device=dev)
        dw = torch.empty(N, K, device=dev)
        wt = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
        wws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
        r0 = r1 = denom = keep = inv = None
        split = 0
        if masked:
            r0 = (torch.rand(M, device=dev) > 0.05).float()
            r1 = torch.ones(M, device=dev)
            denom = torch.full((M,), float(K), device=dev)
            inv = 1.0 / denom
            split = (K // 2 // 4) * 4
        nb = L.tsii_pw_bwd_dw_ws_bytes(M, N, K)
        ws = torch.empty(nb // 4 + 4, device=dev)
        fl = 2.0 * M * K * N
        res = []
        if args.only in ("", "fwd", "nt"):
            t = timeit(lambda: call("tsii_pw_fwd", ptr(x), M, K, ptr(w), N, None, ptr(r0), split, ptr(r1), ptr(denom), None, ptr(y), ptr(wws), wws.numel() * 4, st))
            res.append(f"fwd {t:7.3f} ms {fl / t / 1e9:6.1f} TF/s {4.0 * M * (K + N) / t / 1e9:5.2f} TB/s")
        if args.only in ("", "fwdbn", "nt"):      # forward with the producer's BatchNorm + LeakyReLU applied on load and the statistics epilogue
            sc = torch.rand(K, device=dev) + 0.5
            sh = torch.randn(K, device=dev)
            part = torch.empty(L.tsii_pw_stat_rows(M), 4, N, device=dev)
            t = timeit(lambda: call("tsii_pw_fwd_bn", ptr(x), M, K, ptr(w), N, None, ptr(r0), split, ptr(r1), ptr(denom), None, ptr(sc), ptr(sh), 2, 0.3,
                                    ptr(part), ptr(y), ptr(wws), wws.numel() * 4, st))
            res.append(f"fwdbn {t:7.3f} ms {fl / t / 1e9:6.1f} TF/s {4.0 * M * (K + N) / t / 1e9:5.2f} TB/s")
        if args.only in ("", "dx", "nt"):
            t = timeit(lambda: call("tsii_pw_bwd_dx", ptr(dy), M, N, ptr(w), K, ptr(inv), ptr(r0), split, ptr(r1), ptr(dx), ptr(wt), st))
            res.append(f"dx {t:7.3f} ms {fl / t / 1e9:6.1f} TF/s {4.0 * M * (K + N) / t / 1e9:5.2f} TB/s")
        if args.only in ("", "dxbn", "nt"):   # dX with the K6c BatchNorm-backward reductions in the epilogue
            mean = torch.zeros(K, device=dev); var = torch.ones(K, device=dev)
            gamma = torch.ones(K, device=dev); beta = torch.zeros(K, device=dev)
            bpart = torch.empty(L.tsii_pw_stat_rows(M), 2, K, device=dev)
            t = timeit(lambda: call("tsii_pw_bwd_dx_bn", ptr(dy), M, N, ptr(w), K, ptr(inv), ptr(r0), split, ptr(r1), ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                    1e-5, 2, 0.3, ptr(dx), ptr(bpart), ptr(wt), st))
            res.append(f"dxbn {t:7.3f} ms {fl / t / 1e9:6.1f} TF/s {4.0 * M * (2 * K + N) / t / 1e9:5.2f} TB/s")
        if args.only in ("", "dw"):
            t = timeit(lambda: call("tsii_pw_bwd_dw", ptr(dy), ptr(x), M, N, K, ptr(inv), None, ptr(r0), split, ptr(r1), ptr(dw), None, ptr(ws), nb, st))
            res.append(f"dw {t:7.3f} ms {fl / t / 1e9:6.1f} TF/s")
        gb = 4.0 * M * (K + N) / 1e9
        print(f"mode={gm} M={M:8d} K={K:5d} N={N:5d} masked={int(masked)} alg {gb:5.2f} GB | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()

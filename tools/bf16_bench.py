#!/usr/bin/env python3
"""Micro-benchmark of the bf16-storage entry points on XceptionTextSegment's layer shapes (cfg 5: 1024^2, 8 images).
    python tools/bf16_bench.py [--iters 10] [--only pw,dw,dense,bn]
Prints per entry point: microseconds, algorithmic TB/s (2 bytes per activation element), TFLOP/s for the matrix products."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

BF16 = torch.bfloat16
PW = [(131072, 512, 512), (131072, 256, 512), (524288, 128, 128), (524288, 128, 256), (2097152, 64, 128), (131072, 1024, 256), (524288, 128, 48)]
DW = [(8, 128, 128, 512, 1, 2), (8, 128, 128, 512, 1, 4), (8, 128, 128, 512, 1, 1), (8, 256, 256, 128, 1, 1), (8, 512, 512, 64, 1, 1), (8, 512, 512, 128, 2, 1), (8, 256, 256, 256, 2, 1)]
DENSE = [(8, 128, 128, 512, 256, 3, 1, 3), (8, 128, 128, 512, 256, 3, 1, 1), (8, 256, 256, 304, 128, 3, 1, 1), (8, 512, 512, 32, 64, 3, 1, 1), (8, 256, 256, 128, 256, 1, 2, 1)]
BN = [(131072, 512), (524288, 128), (2097152, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="pw,dw,dense,bn")
    ap.add_argument("--big", action="store_true", help="add the plain 65536 x 4096 x 4096 product (the matrix-core microbench)")
    ap.add_argument("--gemm", default="", help="M,K,N[;M,K,N...]: time only the plain forward product of these shapes, with operands "
                    "drawn as --fill says (the chip is power-capped: the sustained matrix rate depends on how many operand bits toggle)")
    ap.add_argument("--fill", default="normal", choices=["normal", "uniform", "zeros"])
    ap.add_argument("--pw-shapes", default="", help="M,K,N[;...]: replace the 1x1 shape list")
    ap.add_argument("--split-fused", action="store_true", help="1x1 forward: also time the statistics epilogue and the BatchNorm-on-load alone")
    args = ap.parse_args()
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    only = set(args.only.split(","))

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3      # us

    def f32(*shape, scale=1.0):
        return torch.randn(*shape, device=dev) * scale

    def bf(*shape):
        return torch.randn(*shape, device=dev).to(BF16)

    def ws(nbytes):
        return torch.empty(max(4, (int(nbytes) + 3) // 4), dtype=torch.float32, device=dev)

    def line(what, us, nbytes, flops=0.0):
        s = f"  {what:34s} {us:9.1f} us  {nbytes / us / 1e6:6.2f} TB/s"
        if flops:
            s += f"  {flops / us / 1e6:8.1f} TF/s"
        print(s, flush=True)

    if args.gemm:
        for spec in args.gemm.split(";"):
            M, K, N = (int(v) for v in spec.split(","))
            if args.fill == "zeros":
                x, w = torch.zeros(M, K, dtype=BF16, device=dev), torch.zeros(N, K, device=dev)
            elif args.fill == "uniform":
                x, w = (torch.rand(M, K, device=dev) * 2 - 1).to(BF16), torch.rand(N, K, device=dev) * 2 - 1
            else:
                x, w = bf(M, K), f32(N, K, scale=0.05)
            y = torch.empty(M, N, dtype=BF16, device=dev)
            wb = L.tsii_bf16_pw_ws_bytes(N, K)
            w1 = ws(wb)
            print(f"gemm  M={M} K={K} N={N} fill={args.fill}")
            for rep in range(3):
                line(f"fwd plain, {args.iters} launches", timeit(lambda: call("tsii_bf16_pw_fwd", ptr(x), M, K, ptr(w), N, None, None, None, 0, 0.0, None, ptr(y), ptr(w1), wb, st)),
                     2.0 * M * (K + N), 2.0 * M * K * N)
        return
    if "pw" in only:
        pw_list = [tuple(int(v) for v in t.split(",")) for t in args.pw_shapes.split(";")] if args.pw_shapes else (([(65536, 4096, 4096), (16384, 8192, 8192)] if args.big else []) + PW)
        for M, K, N in pw_list:
            print(f"1x1  M={M} K={K} N={N}")
            x, dy, y, dx = bf(M, K), bf(M, N), torch.empty(M, N, dtype=BF16, device=dev), torch.empty(M, K, dtype=BF16, device=dev)
            w = f32(N, K, scale=0.05)
            sc, sh = torch.rand(K, device=dev) + 0.5, f32(K)
            mean, var, gamma, beta = f32(K), torch.rand(K, device=dev) + 0.5, torch.rand(K, device=dev) + 0.5, f32(K)
            part = torch.empty(int(L.tsii_bf16_stat_rows(M)), 4, N, device=dev)
            bpart = torch.empty(int(L.tsii_bf16_stat_rows(M)), 2, K, device=dev)
            wb = L.tsii_bf16_pw_ws_bytes(N, K)
            w1 = ws(wb)
            dw = torch.empty(N, K, device=dev)
            nb = L.tsii_bf16_pw_bwd_dw_ws_bytes(M, N, K)
            w2 = ws(nb)
            byt, fl = 2.0 * M * (K + N), 2.0 * M * K * N
            line("fwd plain", timeit(lambda: call("tsii_bf16_pw_fwd", ptr(x), M, K, ptr(w), N, None, None, None, 0, 0.0, None, ptr(y), ptr(w1), wb, st)), byt, fl)
            line("fwd BN-on-load + stats", timeit(lambda: call("tsii_bf16_pw_fwd", ptr(x), M, K, ptr(w), N, None, ptr(sc), ptr(sh), 2, 0.3, ptr(part), ptr(y), ptr(w1), wb, st)), byt, fl)
            if args.split_fused:
                line("fwd plain + stats", timeit(lambda: call("tsii_bf16_pw_fwd", ptr(x), M, K, ptr(w), N, None, None, None, 0, 0.0, ptr(part), ptr(y), ptr(w1), wb, st)), byt, fl)
                line("fwd BN-on-load, no stats", timeit(lambda: call("tsii_bf16_pw_fwd", ptr(x), M, K, ptr(w), N, None, ptr(sc), ptr(sh), 2, 0.3, None, ptr(y), ptr(w1), wb, st)), byt, fl)
            line("dX plain", timeit(lambda: call("tsii_bf16_pw_bwd_dx", ptr(dy), M, N, ptr(w), K, None, None, None, None, None, 0.0, 0, 0.0, ptr(dx), None, ptr(w1), wb, st)), byt, fl)
            line("dX + K6c", timeit(lambda: call("tsii_bf16_pw_bwd_dx", ptr(dy), M, N, ptr(w), K, ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), 1e-5, 2, 0.3, ptr(dx), ptr(bpart), ptr(w1), wb, st)), byt + 2.0 * M * K, fl)
            line("dW plain", timeit(lambda: call("tsii_bf16_pw_bwd_dw", ptr(dy), ptr(x), M, N, K, None, None, 0, 0.0, ptr(dw), None, ptr(w2), nb, st)), byt, fl)
            line("dW BN-on-load", timeit(lambda: call("tsii_bf16_pw_bwd_dw", ptr(dy), ptr(x), M, N, K, ptr(sc), ptr(sh), 2, 0.3, ptr(dw), None, ptr(w2), nb, st)), byt, fl)
    if "dw" in only:
        for n, h, wd, c, s, d in DW:
            g = (3, 3, s, s, d, d, d, d)
            ho, wo = (h + 2 * d - 2 * d - 1) // s + 1, (wd + 2 * d - 2 * d - 1) // s + 1
            print(f"dw3x3  [{n},{h},{wd},{c}] stride {s} dilation {d}")
            x, dy = bf(n, h, wd, c), bf(n, ho, wo, c)
            y, dx = torch.empty(n, ho, wo, c, dtype=BF16, device=dev), torch.empty(n, h, wd, c, dtype=BF16, device=dev)
            w = f32(c, 1, 3, 3, scale=0.3)
            sc, sh = torch.rand(c, device=dev) + 0.5, f32(c)
            mean, var, gamma, beta = f32(c), torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) + 0.5, f32(c)
            rows = int(L.tsii_bf16_dw_stat_rows(n, ho, wo, c, 3, 3, s, s, d, d))
            part = torch.empty(rows, 4, c, device=dev)
            brows = int(L.tsii_bf16_dw_bwd_stat_rows(n, h, wd, c, *g))
            bpart = torch.empty(max(brows, 1), 2, c, device=dev)
            nb = L.tsii_bf16_dw_bwd_dw_ws_bytes(n, ho, wo, c, 3, 3, s, s, d, d)
            w2 = ws(nb)
            dwg = torch.empty(c, 1, 3, 3, device=dev)
            byt = 2.0 * n * c * (h * wd + ho * wo)
            line("fwd plain", timeit(lambda: call("tsii_bf16_dw_fwd", ptr(x), ptr(w), None, n, h, wd, c, *g, ho, wo, None, None, 0, 0.0, None, ptr(y), st)), byt)
            line("fwd BN-on-load + stats", timeit(lambda: call("tsii_bf16_dw_fwd", ptr(x), ptr(w), None, n, h, wd, c, *g, ho, wo, ptr(sc), ptr(sh), 2, 0.3, ptr(part), ptr(y), st)), byt)
            line("dX plain", timeit(lambda: call("tsii_bf16_dw_bwd_dx", ptr(dy), ptr(w), n, h, wd, c, *g, ho, wo, None, None, None, None, None, 0.0, 0, 0.0, ptr(dx), None, st)), byt)
            if brows > 0:
                line("dX + K6c", timeit(lambda: call("tsii_bf16_dw_bwd_dx", ptr(dy), ptr(w), n, h, wd, c, *g, ho, wo, ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), 1e-5, 2, 0.3, ptr(dx), ptr(bpart), st)), byt + 2.0 * n * c * h * wd)
            line("dW BN-on-load", timeit(lambda: call("tsii_bf16_dw_bwd_dw", ptr(dy), ptr(x), n, h, wd, c, *g, ho, wo, ptr(sc), ptr(sh), 2, 0.3, ptr(dwg), None, ptr(w2), nb, st)), byt)
        for k in (3, 5, 9):
            n, h, wd, c = 8, 128, 128, 512
            x, y = bf(n, h, wd, c), torch.empty(n, h, wd, c, dtype=BF16, device=dev)
            line(f"avgpool {k} [{n},{h},{wd},{c}]", timeit(lambda: call("tsii_bf16_avgpool", ptr(x), n, h, wd, c, k, ptr(y), st)), 4.0 * n * h * wd * c)
    if "dense" in only:
        for n, h, wd, cin, cout, k, s, d in DENSE:
            p = d * (k - 1) // 2
            g = (k, k, s, s, p, p, d, d)
            ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (wd + 2 * p - d * (k - 1) - 1) // s + 1
            print(f"dense {k}x{k}  [{n},{h},{wd},{cin}] -> {cout} stride {s} dilation {d}")
            x, dy = bf(n, h, wd, cin), bf(n, ho, wo, cout)
            y, dx = torch.empty(n, ho, wo, cout, dtype=BF16, device=dev), torch.empty(n, h, wd, cin, dtype=BF16, device=dev)
            w = f32(cout, cin, k, k, scale=0.05)
            part = torch.empty(int(L.tsii_bf16_stat_rows(n * ho * wo)), 4, cout, device=dev)
            wb = L.tsii_bf16_dense_ws_bytes(cin, cout, k, k)
            w1 = ws(wb)
            nb = L.tsii_bf16_dense_bwd_dw_ws_bytes(n, ho, wo, cin, cout, k, k)
            w2 = ws(nb)
            dwg = torch.empty(cout, cin, k, k, device=dev)
            byt, fl = 2.0 * n * (h * wd * cin + ho * wo * cout), 2.0 * n * ho * wo * cout * cin * k * k
            line("fwd + stats", timeit(lambda: call("tsii_bf16_dense_fwd", ptr(x), ptr(w), None, n, h, wd, cin, cout, *g, ho, wo, ptr(part), ptr(y), ptr(w1), wb, st)), byt, fl)
            line("dX", timeit(lambda: call("tsii_bf16_dense_bwd_dx", ptr(dy), ptr(w), n, h, wd, cin, cout, *g, ho, wo, ptr(dx), ptr(w1), wb, st)), byt, fl)
            line("dW", timeit(lambda: call("tsii_bf16_dense_bwd_dw", ptr(dy), ptr(x), n, h, wd, cin, cout, *g, ho, wo, ptr(dwg), None, ptr(w2), nb, st)), byt, fl)
    if "bn" in only:
        for M, C in BN:
            print(f"BatchNorm  [{M},{C}]")
            y, dout, out = bf(M, C), bf(M, C), torch.empty(M, C, dtype=BF16, device=dev)
            sc, sh = torch.rand(C, device=dev) + 0.5, f32(C)
            mean, var, gamma, beta = f32(C), torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, f32(C)
            rows = int(L.tsii_bf16_bn_stat_rows(M, C))
            part = torch.empty(rows, 4, C, device=dev)
            dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
            nb = L.tsii_bf16_bn_ws_bytes(M, C)
            w1 = ws(nb)
            prow = int(L.tsii_bf16_stat_rows(M))
            bpart = torch.randn(prow, 2, C, device=dev)
            line("stats", timeit(lambda: call("tsii_bf16_bn_stats", ptr(y), M, C, ptr(part), st)), 2.0 * M * C)
            line("apply (+ residual)", timeit(lambda: call("tsii_bf16_bn_act_fwd", ptr(y), M, C, ptr(sc), ptr(sh), 2, 0.3, ptr(dout), ptr(out), st)), 6.0 * M * C)
            line("backward, own reductions", timeit(lambda: call("tsii_bf16_bn_act_bwd", ptr(dout), ptr(y), M, C, ptr(mean), ptr(var), ptr(gamma), ptr(beta), 1e-5, 2, 0.3, 1, None, 0, ptr(out), ptr(dg), ptr(db), ptr(w1), nb, st)), 10.0 * M * C)
            line(f"backward, {prow} partial rows given", timeit(lambda: call("tsii_bf16_bn_act_bwd", ptr(dout), ptr(y), M, C, ptr(mean), ptr(var), ptr(gamma), ptr(beta), 1e-5, 2, 0.3, 1, ptr(bpart), prow, ptr(out), ptr(dg), ptr(db), ptr(w1), nb, st)), 6.0 * M * C)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""On-chip check of the producer / consumer GEMM (gemm_pc.hip): forward with BatchNorm-on-load + statistics, plain forward,
dX and dX with the BatchNorm-backward reductions on ImageFill's layer shapes against fp64 torch, several runs each
(bitwise repeatability = no LDS race).    python tools/pc_check.py [--runs 4]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = [(65536, 192, 384), (32768, 64, 256), (16384, 384, 768), (8192, 256, 1024), (8192, 1024, 256), (4099, 40, 200),
          (20480, 128, 512), (16384, 512, 128), (16384, 384, 64), (69632, 96, 192), (131072, 32, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=4)
    args = ap.parse_args()
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    worst = 0.0
    for (M, K, N) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(M + K + N)
        x = torch.randn(M, K, device=dev, generator=g) * 2 + 0.5
        w = torch.randn(N, K, device=dev, generator=g) * 0.2
        b = torch.randn(N, device=dev, generator=g)
        sc = torch.rand(K, device=dev, generator=g) + 0.5
        sh = torch.randn(K, device=dev, generator=g)
        r0 = (torch.rand(M, device=dev, generator=g) > 0.05).float()
        r1 = (torch.rand(M, device=dev, generator=g) > 0.05).float()
        denom = torch.randint(1, 9, (M,), device=dev, generator=g).float()
        keep = (torch.rand(M, device=dev, generator=g) > 0.1).float()
        split = (K // 2 // 4) * 4
        wws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
        rows = L.tsii_pw_stat_rows(M)
        z = x.double() * sc.double() + sh.double()
        a = torch.where(z > 0, z, 0.3 * z)
        a[:, :split] *= r0.double()[:, None]
        a[:, split:] *= r1.double()[:, None]
        ref = (a @ w.double().t() / denom.double()[:, None] + b.double()) * keep.double()[:, None]
        outs = []
        for _ in range(args.runs):
            y = torch.full((M, N), float("nan"), device=dev)
            part = torch.zeros(rows, 4, N, device=dev)
            call("tsii_pw_fwd_bn", ptr(x), M, K, ptr(w), N, ptr(b), ptr(r0), split, ptr(r1), ptr(denom), ptr(keep), ptr(sc), ptr(sh), 2, 0.3,
                 ptr(part), ptr(y), ptr(wws), wws.numel() * 4, st)
            torch.cuda.synchronize()
            outs.append((y, part))
        e_f = float((outs[0][0].double() - ref).abs().max() / ref.abs().max())
        rep_f = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        y64 = outs[0][0].double()
        part = outs[0][1].double()
        # statistics: sum over blocks of (n, pivot, s1, s2) -> mean / E[y^2]
        mean = ((part[:, 1] * part[:, 0] + part[:, 2]).sum(0)) / M
        e_s = float((mean - y64.mean(0)).abs().max() / (y64.abs().max() + 1e-30))
        # dX (+ K6c reductions)
        dy = torch.randn(M, N, device=dev, generator=g)
        inv = keep / denom
        gg = dy.double() * inv.double()[:, None]
        rdx = gg @ w.double()
        rdx[:, :split] *= r0.double()[:, None]
        rdx[:, split:] *= r1.double()[:, None]
        wt = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
        mean_x = x.double().mean(0).float()
        var_x = x.double().var(0, unbiased=False).float()
        gamma = torch.rand(K, device=dev, generator=g) + 0.5
        beta = torch.randn(K, device=dev, generator=g)
        outs = []
        for _ in range(args.runs):
            dx = torch.full((M, K), float("nan"), device=dev)
            bpart = torch.zeros(rows, 2, K, device=dev)
            call("tsii_pw_bwd_dx_bn", ptr(dy), M, N, ptr(w), K, ptr(inv), ptr(r0), split, ptr(r1), ptr(x), ptr(mean_x), ptr(var_x), ptr(gamma), ptr(beta),
                 1e-5, 2, 0.3, ptr(dx), ptr(bpart), ptr(wt), st)
            torch.cuda.synchronize()
            outs.append((dx, bpart))
        e_d = float((outs[0][0].double() - rdx).abs().max() / rdx.abs().max())
        rep_d = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        xh = (x.double() - mean_x.double()) / torch.sqrt(var_x.double() + 1e-5)
        zz = xh * gamma.double() + beta.double()
        dz = outs[0][0].double() * torch.where(zz > 0, 1.0, 0.3)
        s1 = outs[0][1][:, 0].double().sum(0)
        e_b = float((s1 - dz.sum(0)).abs().max() / dz.abs().sum(0).max())
        dx2 = torch.full((M, K), float("nan"), device=dev)
        call("tsii_pw_bwd_dx", ptr(dy), M, N, ptr(w), K, ptr(inv), ptr(r0), split, ptr(r1), ptr(dx2), ptr(wt), st)
        torch.cuda.synchronize()
        e_d2 = float((dx2.double() - rdx).abs().max() / rdx.abs().max())
        worst = max(worst, e_f, e_d, e_d2, e_s, e_b)
        print(f"M={M:6d} K={K:5d} N={N:5d}  fwd_bn err {e_f:.2e} (stats {e_s:.2e}) repeat {rep_f} | dx_bn err {e_d:.2e} (sums {e_b:.2e}) repeat {rep_d} | dx err {e_d2:.2e}",
              flush=True)
        assert rep_f and rep_d, "results differ between runs: LDS race?"
        assert max(e_f, e_d, e_d2) < 2e-6 and e_s < 1e-5 and e_b < 1e-4, "accuracy"
    print(f"pc_check ok, worst {worst:.2e}")


if __name__ == "__main__":
    main()

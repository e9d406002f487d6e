#!/usr/bin/env python3
"""The 1x1-convolution forward / dX launches (class gemm_nt) of ONE ImageFill 512^2 bs-32 training step, replayed one by one.

    python tools/nt_bench.py [--iters 10] [--batch 32] [--size 512] [--only fwd_up,dx_bn] [--min-ms 0.0]

Step 1 records every tsii_pw_{fwd,fwd_bn,fwd_up,bwd_dx,bwd_dx_bn} call of a real step (scalar arguments + which pointers were
non-NULL); step 2 replays each distinct launch on fresh random buffers of the same sizes under HIP events and prints its time, the
algorithmic bytes / HBM floor at 8 TB/s, the 6-product MFMA floor, and the count-weighted total per step.  With
TSII_LIBRARY=<variant .so> the same list runs on an A/B build (tools/variants/build_variant.py)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

NAMES = ("tsii_pw_fwd", "tsii_pw_fwd_bn", "tsii_pw_fwd_up", "tsii_pw_bwd_dx", "tsii_pw_bwd_dx_bn")
NAMES_TN = ("tsii_pw_bwd_dw", "tsii_pw_bwd_dw_bn")      # --tn: the weight-gradient launches (class gemm_tn) instead


def record(batch, size, names=NAMES):
    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = T.ImageFill().to(dev).train()
    tr = FlatSGDTrainer(model, lr=1e-3, momentum=0.9, weight_decay=1e-4)
    corrupted, mask, clean = make_batch(batch, size, seed0=0)
    corrupted, mask, clean = corrupted.to(dev), mask.to(dev), to_nhwc(clean.to(dev))
    tr.step(corrupted, mask, clean)
    calls, real = [], _lib.call

    def spy(name, *args):
        if name in names:
            calls.append((name, tuple((a if isinstance(a, (int, float)) else (a is not None and getattr(a, "value", 1) is not None)) for a in args)))
        return real(name, *args)
    _lib.call = spy
    import text_segmentation_image_inpainting_amd.ops as ops
    ops.call = spy
    try:
        tr.step(corrupted, mask, clean)
    finally:
        _lib.call = ops.call = real
    torch.cuda.synchronize()
    del tr, model
    torch.cuda.empty_cache()
    uniq = {}
    for c in calls:
        uniq[c] = uniq.get(c, 0) + 1
    return uniq


def replay(name, a, iters):
    """-> (ms, algorithmic bytes, MACs, label)"""
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    R = lambda *s: torch.randn(*s, device=dev)
    opt = lambda on, t: t if on else None
    if name in NAMES_TN:
        # (gy, x, m, cout, k, inv, keep, r0, split, r1, [in_scale, in_shift, act, slope,] dw, db, ws, nbytes, stream)
        m, cout, k = a[2], a[3], a[4]
        gy, x = R(m, cout), R(m, k)
        inv = opt(a[5], torch.full((m,), 1.0 / cout, device=dev)); keep = opt(a[6], torch.ones(m, device=dev))
        r0 = opt(a[7], (torch.rand(m, device=dev) > 0.05).float()); split = a[8]; r1 = opt(a[9], torch.ones(m, device=dev))
        bn = name.endswith("_bn")
        o = 4 if bn else 0
        dw = torch.empty(cout, k, device=dev); db = opt(a[11 + o], torch.empty(cout, device=dev))
        nb = L.tsii_pw_bwd_dw_ws_bytes(m, cout, k)
        ws = torch.empty(nb // 4 + 4, device=dev)
        if bn:
            sc, sh = torch.rand(k, device=dev) + 0.5, R(k)
            fn = lambda: call(name, ptr(gy), ptr(x), m, cout, k, ptr(inv), ptr(keep), ptr(r0), split, ptr(r1), ptr(sc), ptr(sh), a[12], a[13], ptr(dw), ptr(db), ptr(ws), nb, st)
        else:
            fn = lambda: call(name, ptr(gy), ptr(x), m, cout, k, ptr(inv), ptr(keep), ptr(r0), split, ptr(r1), ptr(dw), ptr(db), ptr(ws), nb, st)
        by, label, n = 4.0 * m * (k + cout), f"P={cout:4d} Q={k:4d}" + (" inBN" if bn else ""), cout
    elif name in ("tsii_pw_fwd", "tsii_pw_fwd_bn", "tsii_pw_fwd_up"):
        m, k, n = a[1], a[2], a[4]
        x, w, y = R(m, k), R(n, k) * 0.05, torch.empty(m, n, device=dev)
        bias = opt(a[5], R(n))
        r0 = opt(a[6], (torch.rand(m, device=dev) > 0.05).float()); split = a[7]; r1 = opt(a[8], torch.ones(m, device=dev))
        denom = opt(a[9], torch.full((m,), float(k), device=dev)); keep = opt(a[10], torch.ones(m, device=dev))
        wb = L.tsii_pw_ws_bytes(n, k)
        ws = torch.empty(wb // 4 + 4, device=dev)
        by, label = 4.0 * m * (k + n), f"K={k:4d} N={n:4d}"
        if name == "tsii_pw_fwd":
            fn = lambda: call(name, ptr(x), m, k, ptr(w), n, ptr(bias), ptr(r0), split, ptr(r1), ptr(denom), ptr(keep), ptr(y), ptr(ws), wb, st)
        elif name == "tsii_pw_fwd_bn":
            sc, sh = opt(a[11], torch.rand(k, device=dev) + 0.5), opt(a[12], R(k))
            part = opt(a[15], torch.empty(L.tsii_pw_stat_rows(m), 4, n, device=dev))
            fn = lambda: call(name, ptr(x), m, k, ptr(w), n, ptr(bias), ptr(r0), split, ptr(r1), ptr(denom), ptr(keep), ptr(sc), ptr(sh), a[13], a[14],
                              ptr(part), ptr(y), ptr(ws), wb, st)
            label += (" inBN" if a[11] else "") + (" stats" if a[15] else "")
        else:
            h, wd = a[12], a[13]
            up = R(m // 4, n)
            part = opt(a[14], torch.empty(L.tsii_pw_stat_rows(m), 4, n, device=dev))
            fn = lambda: call(name, ptr(x), m, k, ptr(w), n, ptr(bias), ptr(r0), split, ptr(r1), ptr(denom), ptr(keep), ptr(up), h, wd,
                              ptr(part), ptr(y), ptr(ws), wb, st)
            by += 1.0 * m * n
            label += f" up{h}x{wd}" + (" stats" if a[14] else "")
    else:
        m, cout, k = a[1], a[2], a[4]
        gy, w, dx = R(m, cout), R(cout, k) * 0.05, torch.empty(m, k, device=dev)
        inv = opt(a[5], torch.full((m,), 1.0 / cout, device=dev))
        r0 = opt(a[6], (torch.rand(m, device=dev) > 0.05).float()); split = a[7]; r1 = opt(a[8], torch.ones(m, device=dev))
        wt = torch.empty(L.tsii_pw_ws_bytes(cout, k) // 4 + 4, device=dev)
        by, label = 4.0 * m * (k + cout), f"K={cout:4d} N={k:4d}"
        if name == "tsii_pw_bwd_dx":
            fn = lambda: call(name, ptr(gy), m, cout, ptr(w), k, ptr(inv), ptr(r0), split, ptr(r1), ptr(dx), ptr(wt), st)
        else:
            x = R(m, k)
            mean, var, gamma, beta = torch.zeros(k, device=dev), torch.ones(k, device=dev), torch.ones(k, device=dev), torch.zeros(k, device=dev)
            part = torch.empty(L.tsii_pw_stat_rows(m), 2, k, device=dev)
            fn = lambda: call(name, ptr(gy), m, cout, ptr(w), k, ptr(inv), ptr(r0), split, ptr(r1), ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                              a[14], a[15], a[16], ptr(dx), ptr(part), ptr(wt), st)
            by += 4.0 * m * k
            label += " K6c"
        n = cout
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, by, float(m) * k * n, f"M={m:8d} " + label


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--only", default="", help="comma list of substrings of the entry-point names")
    ap.add_argument("--tn", action="store_true", help="the weight-gradient launches (tsii_pw_bwd_dw / _dw_bn) instead of forward / dX")
    ap.add_argument("--model", default="ImageFill")
    ap.add_argument("--min-k", type=int, default=0)
    ap.add_argument("--max-k", type=int, default=1 << 30, help="reduction length (the entry point's contraction dimension)")
    args = ap.parse_args()
    uniq = record(args.batch, args.size, NAMES_TN if args.tn else NAMES)
    rows, tot, tot_floor = [], 0.0, 0.0
    for (name, a), cnt in uniq.items():
        if args.only and not any(s in name for s in args.only.split(",")):
            continue
        kk = a[4] if name in NAMES_TN else a[2]
        if not (args.min_k <= kk <= args.max_k):
            continue
        ms, by, macs, label = replay(name, a, args.iters)
        t_hbm, t_mfma = by / 8e12 * 1e3, 2.0 * macs * 6 / 2.5e15 * 1e3
        rows.append((ms * cnt, name[8:], label, cnt, ms, by, t_hbm, t_mfma))
        tot += ms * cnt
        tot_floor += max(t_hbm, t_mfma) * cnt
    rows.sort(reverse=True)
    print(f"library: {os.environ.get('TSII_LIBRARY', 'stock')}")
    for tms, nm, label, cnt, ms, by, t_hbm, t_mfma in rows:
        print(f"{nm:10s} x{cnt:<2d} {label:44s} {ms:7.3f} ms  {by / ms / 1e9:5.2f} TB/s  floor hbm {t_hbm:.3f} mfma {t_mfma:.3f}  frac {max(t_hbm, t_mfma) / ms:.2f}", flush=True)
    print(f"TOTAL per step: {tot:.3f} ms over {sum(r[3] for r in rows)} launches; sum of per-launch floors {tot_floor:.3f} ms; per-launch roofline frac {tot_floor / tot:.3f}")


if __name__ == "__main__":
    main()

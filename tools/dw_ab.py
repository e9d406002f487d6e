#!/usr/bin/env python3
"""A/B of the depth-wise entry points between two builds of the library on the shapes a network really calls them with:
records the scalar arguments of every tsii_dw_* call of one training step of --model, then replays each distinct call on random
operands through BOTH libraries and compares the outputs (max |a-b| / max |b|; statistics after tsii_bn_finalize; K6c partials summed).
    python tools/dw_ab.py --other tools/variants/_bin/libtsii_nolean.so [--model TextSegament --size 256 --batch 2]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--other", required=True)
    ap.add_argument("--model", default="TextSegament")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd import _lib, ops
    from text_segmentation_image_inpainting_amd._lib import ptr
    dev = torch.device("cuda:0")
    A = _lib.lib()
    B = _lib.bind(ctypes.CDLL(args.other))
    st = _lib.stream()
    seen = {}
    real = ops.call
    def rec(name, *a):
        if name.startswith("tsii_dw_"):
            seen.setdefault((name, tuple(x for x in a if isinstance(x, (int, float))), tuple(x is None for x in a)), 0)
        return real(name, *a)
    ops.call = rec
    torch.manual_seed(0)
    if args.model == "ImageFill":
        from text_segmentation_image_inpainting_amd.synthetic import make_batch
        m = T.ImageFill().to(dev).train()
        c, mk, cl = make_batch(args.batch, args.size)
        m((c.to(dev), mk.to(dev))).sum().backward()
    else:
        m = getattr(T, args.model)().to(dev).train()
        m(torch.randn(args.batch, 3, args.size, args.size, device=dev)).sum().backward()
    ops.call = real
    torch.cuda.synchronize()
    worst = 0.0
    for (name, sc, nones) in seen:
        g = torch.Generator(device=dev); g.manual_seed(1)
        rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
        if name in ("tsii_dw_fwd", "tsii_dw_fwd_bn"):
            n, h, w, c = sc[0:4]; geom = sc[4:12]; ho, wo = sc[12:14]
            x = rnd(n, h, w, c) + 0.3; wt = rnd(c, 1, geom[0], geom[1])
            # argument order: x, rmask, w, bias, denom, keep, n,h,w,c, geom(8), ho, wo, [sc, sh, act, slope, part], y, ws, stream
            rmask = None if nones[1] else (torch.rand(n, h, w, device=dev, generator=g) > 0.2).float()
            bias = None if nones[3] else rnd(c)
            denom = None if nones[4] else torch.full((n, ho, wo), 9.0 * c, device=dev)
            keep = None if nones[5] else (torch.rand(n, ho, wo, device=dev, generator=g) > 0.1).float()
            outs = []
            if name == "tsii_dw_fwd_bn":
                has_bn = not nones[20]
                s_ = (torch.rand(c, device=dev, generator=g) + 0.5) if has_bn else None
                h_ = rnd(c) if has_bn else None
            for L in (A, B):
                y = torch.full((n, ho, wo, c), float("nan"), device=dev); ws = torch.empty(c * 9 + 16, device=dev)
                if name == "tsii_dw_fwd":
                    rc = L.tsii_dw_fwd(ptr(x), ptr(rmask), ptr(wt), ptr(bias), ptr(denom), ptr(keep), n, h, w, c, *geom, ho, wo, ptr(y), ptr(ws), st)
                    outs.append((rc, y, None))
                else:
                    act, slope = int(sc[14]), float(sc[15])
                    rows = L.tsii_dw_stat_rows(n, ho, wo, c, geom[0], geom[1], geom[2], geom[3], geom[6], geom[7])
                    part = torch.zeros(max(rows, 1) * 4 * c, device=dev) if not nones[24] else None
                    rc = L.tsii_dw_fwd_bn(ptr(x), ptr(rmask), ptr(wt), ptr(bias), ptr(denom), ptr(keep), n, h, w, c, *geom, ho, wo, ptr(s_), ptr(h_), act, slope,
                                          ptr(part), ptr(y), ptr(ws), st)
                    stat = None
                    if part is not None:
                        mean = torch.empty(c, device=dev); var = torch.empty(c, device=dev)
                        nb = L.tsii_bn_finalize_ws_bytes(rows, c); w2 = torch.empty(nb // 4 + 4, device=dev)
                        L.tsii_bn_finalize(ptr(part), rows, c, n * ho * wo, ptr(mean), ptr(var), None, None, 0.1, None, None, 1e-5, None, None, ptr(w2), nb, st)
                        stat = torch.stack([mean, var])
                    outs.append((rc, y, stat))
            torch.cuda.synchronize()
            (ra, ya, sa), (rb, yb, sb) = outs
            e = float((ya - yb).abs().max() / yb.abs().max().clamp_min(1e-30))
            es = float(((sa - sb).abs() / sb.abs().clamp_min(1e-3)).max()) if sa is not None else 0.0
            if sa is not None:      # which of the two is right: fp64 statistics of the (identical) outputs
                y64 = yb.double().reshape(-1, c)
                ref = torch.stack([y64.mean(0), y64.var(0, unbiased=False)])
                ea = float(((sa.double() - ref).abs() / ref.abs().clamp_min(1e-3)).max()); eb = float(((sb.double() - ref).abs() / ref.abs().clamp_min(1e-3)).max())
                print(f"      statistics vs fp64: this build {ea:.2e}, other build {eb:.2e}")
            nan = bool(torch.isnan(ya).any())
            worst = max(worst, e, es)
            print(f"{name:16s} rc {ra}/{rb} y {e:.2e} stats {es:.2e} nan {nan}  {sc[:16]} none={[i for i, v in enumerate(nones) if v]}", flush=True)
        elif name in ("tsii_dw_bwd_dx", "tsii_dw_bwd_dx_bn"):
            n, h, w, c = sc[0:4]; geom = sc[4:12]; ho, wo = sc[12:14]
            dy = rnd(n, ho, wo, c); wt = rnd(c, 1, geom[0], geom[1])
            inv = None if nones[1] else torch.rand(n, ho, wo, device=dev, generator=g)
            rmask = None if nones[3] else (torch.rand(n, h, w, device=dev, generator=g) > 0.2).float()
            outs = []
            if name == "tsii_dw_bwd_dx_bn":
                yraw = rnd(n, h, w, c); mean = rnd(c); var = torch.rand(c, device=dev, generator=g) + 0.5; gam = torch.rand(c, device=dev, generator=g) + 0.5; bet = rnd(c)
                eps, act, slope = float(sc[14]), int(sc[15]), float(sc[16])
            for L in (A, B):
                dx = torch.full((n, h, w, c), float("nan"), device=dev); ws = torch.empty(c * 9 + 16, device=dev)
                if name == "tsii_dw_bwd_dx":
                    rc = L.tsii_dw_bwd_dx(ptr(dy), ptr(inv), ptr(wt), ptr(rmask), n, h, w, c, *geom, ho, wo, ptr(dx), ptr(ws), st)
                    outs.append((rc, dx, None))
                else:
                    rows = L.tsii_dw_bwd_stat_rows(n, h, w, c, *geom)
                    part = torch.zeros(max(rows, 1) * 2 * c, device=dev)
                    rc = L.tsii_dw_bwd_dx_bn(ptr(dy), ptr(inv), ptr(wt), ptr(rmask), n, h, w, c, *geom, ho, wo, ptr(yraw), ptr(mean), ptr(var), ptr(gam), ptr(bet),
                                             eps, act, slope, ptr(dx), ptr(part), ptr(ws), st)
                    outs.append((rc, dx, part.view(rows, 2, c).double().sum(0)))
            torch.cuda.synchronize()
            (ra, ya, sa), (rb, yb, sb) = outs
            e = float((ya - yb).abs().max() / yb.abs().max().clamp_min(1e-30))
            es = float(((sa - sb).abs() / sb.abs().max()).max()) if sa is not None else 0.0
            worst = max(worst, e, es)
            print(f"{name:16s} rc {ra}/{rb} dx {e:.2e} partials {es:.2e} nan {bool(torch.isnan(ya).any())}  {sc[:17]} none={[i for i, v in enumerate(nones) if v]}", flush=True)
    print("worst", worst)

if __name__ == "__main__":
    main()

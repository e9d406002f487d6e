#!/usr/bin/env python3
"""Micro-benchmark of the depth-wise entry points on ImageFill's largest layers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from text_segmentation_image_inpainting_amd import _lib
from text_segmentation_image_inpainting_amd._lib import call, ptr
L = _lib.lib(); dev = torch.device("cuda:0"); st = _lib.stream()
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
# default: ImageFill's largest layers (mask planes); "cfg3": TextSegament 512^2 bs 64's dilated layers (no mask planes); "dil": ImageFill's
# dilated encoder levels
SETS = {"": [(32, 256, 256, 384, 1, 1), (32, 128, 128, 768, 1, 1), (32, 64, 64, 1024, 1, 1), (32, 256, 256, 256, 2, 1)],
        "cfg3": [(64, 64, 64, 1920, 1, 16), (64, 64, 64, 1920, 1, 8), (64, 64, 64, 1152, 1, 8), (64, 64, 64, 1152, 1, 4), (64, 64, 64, 768, 1, 4), (64, 64, 64, 768, 1, 2), (64, 128, 128, 384, 1, 2), (64, 64, 64, 384, 1, 1)],
        "dil": [(32, 64, 64, 1024, 1, 2), (32, 64, 64, 1024, 1, 4), (32, 64, 64, 1024, 1, 8)]}
WHICH = sys.argv[1] if len(sys.argv) > 1 else ""
MASKED = WHICH != "cfg3"
for (n, h, w, c, s, d) in SETS[WHICH]:
    ho, wo = (h + 2 * d - 2 * d - 1) // s + 1, (w + 2 * d - 2 * d - 1) // s + 1
    x = torch.randn(n, h, w, c, device=dev); wt = torch.randn(c, 1, 3, 3, device=dev)
    m = (torch.rand(n, h, w, device=dev) > 0.05).float() if MASKED else None
    den = torch.full((n, ho, wo), 9.0 * c, device=dev) if MASKED else None
    keep = torch.ones(n, ho, wo, device=dev) if MASKED else None
    inv = 1 / den if MASKED else None
    y = torch.empty(n, ho, wo, c, device=dev); dy = torch.randn(n, ho, wo, c, device=dev); dx = torch.empty_like(x)
    ws = torch.empty(c * 9 + 16, device=dev)
    g = (3, 3, s, s, d, d, d, d)
    tf = timeit(lambda: call("tsii_dw_fwd", ptr(x), ptr(m), ptr(wt), None, ptr(den), ptr(keep), n, h, w, c, *g, ho, wo, ptr(y), ptr(ws), st))
    tb = timeit(lambda: call("tsii_dw_bwd_dx", ptr(dy), ptr(inv), ptr(wt), ptr(m), n, h, w, c, *g, ho, wo, ptr(dx), ptr(ws), st))
    nb = L.tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, 3, 3); w2 = torch.empty(nb // 4 + 4, device=dev); dwg = torch.empty_like(wt)
    tw = timeit(lambda: call("tsii_dw_bwd_dw", ptr(dy), ptr(inv), ptr(keep), ptr(x), ptr(m), n, h, w, c, *g, ho, wo, ptr(dwg), None, ptr(w2), nb, st))
    gb = (x.numel() + y.numel()) * 4 / 1e9
    # the BatchNorm-fused forms (K6b forward: BatchNorm + LeakyReLU on load and statistics partials; K6c dX: reductions of the
    # BatchNorm backward of the tensor it feeds)
    sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev)
    rows = L.tsii_dw_stat_rows(n, ho, wo, c, 3, 3, s, s, d, d)
    line = f"dw n{n} {h}x{w} c{c} s{s} d{d}: fwd {tf:6.3f} ms {gb / tf:5.2f} TB/s | dx {tb:6.3f} ms {gb / tb:5.2f} TB/s | dw {tw:6.3f} ms {gb / tw:5.2f} TB/s"
    if rows > 0:
        part = torch.empty(rows * 4 * c, device=dev)
        tfb = timeit(lambda: call("tsii_dw_fwd_bn", ptr(x), ptr(m), ptr(wt), None, ptr(den), ptr(keep), n, h, w, c, *g, ho, wo, ptr(sc), ptr(sh), 2, 0.3,
                                  ptr(part), ptr(y), ptr(ws), st))
        line += f" | fwd_bn {tfb:6.3f} ms {gb / tfb:5.2f} TB/s"
    brows = L.tsii_dw_bwd_stat_rows(n, h, w, c, 3, 3, s, s, d, d, d, d)
    if brows > 0:
        bpart = torch.empty(brows * 2 * c, device=dev)
        mean = torch.randn(c, device=dev); var = torch.rand(c, device=dev) + 0.5; gam = torch.rand(c, device=dev) + 0.5; bet = torch.randn(c, device=dev)
        tdb = timeit(lambda: call("tsii_dw_bwd_dx_bn", ptr(dy), ptr(inv), ptr(wt), ptr(m), n, h, w, c, *g, ho, wo, ptr(x), ptr(mean), ptr(var), ptr(gam), ptr(bet),
                                  1e-5, 2, 0.3, ptr(dx), ptr(bpart), ptr(ws), st))
        gb3 = gb + x.numel() * 4 / 1e9
        line += f" | dx_bn {tdb:6.3f} ms {gb3 / tdb:5.2f} TB/s"
    print(line, flush=True)

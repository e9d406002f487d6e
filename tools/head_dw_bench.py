#!/usr/bin/env python3
"""Times tsii_head_cat_bwd_dw_low / tsii_head_cat_bwd_dw / tsii_head_cat_fwd on ImageFill's head (32 x 512^2, 32 + 3 -> 3) with
HIP events, per library build (TSII_LIBRARY selects the build; run once per build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from text_segmentation_image_inpainting_amd import _lib, ops
from text_segmentation_image_inpainting_amd.ops import ptr

def main():
    dev = torch.device("cuda:0")
    n, h, w, c1, c2, cout = 32, 512, 512, 32, 3, 3
    L, st = _lib.lib(), _lib.stream()
    g = torch.Generator(device=dev).manual_seed(0)
    low = torch.randn((n, h // 2, w // 2, c1), device=dev, generator=g)
    skip = torch.randn((n, h, w, c2), device=dev, generator=g)
    dy = torch.randn((n, h, w, cout), device=dev, generator=g)
    inv = torch.rand((n, h, w), device=dev, generator=g)
    r0l = (torch.rand((n, h // 2, w // 2), device=dev, generator=g) > 0.1).float()
    r0 = r0l.repeat_interleave(2, 1).repeat_interleave(2, 2).contiguous()
    dw = torch.empty((cout, c1 + c2, 3, 3), device=dev); db = torch.empty(cout, device=dev)
    nb = L.tsii_dense_bwd_dw_ws_bytes(n, h, w, c1 + c2, cout, 3, 3)
    ws = torch.empty(nb // 4 + 64, device=dev)
    def timed(fn, reps=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [fn() for _ in range(reps)]; e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    new = lambda: ops.call("tsii_head_cat_bwd_dw_low", ptr(dy), ptr(inv), None, ptr(low), ptr(skip), c1, c2, ptr(r0l), None, n, h, w, cout, ptr(dw), None, ptr(ws), nb, st)
    old = lambda: ops.call("tsii_head_cat_bwd_dw", ptr(dy), ptr(inv), None, ptr(low), ptr(skip), c1, c2, ptr(r0), None, n, h, w, cout, ptr(dw), None, ptr(ws), nb, st)
    wt = torch.randn((cout, c1 + c2, 3, 3), device=dev, generator=g) * 0.1
    bias = torch.randn(cout, device=dev, generator=g)
    denom = torch.rand((n, h, w), device=dev, generator=g) + 0.5
    keep = (torch.rand((n, h, w), device=dev, generator=g) > 0.05).float()
    yo = torch.empty((n, h, w, cout), device=dev)
    nf = L.tsii_dense_ws_bytes(c1 + c2, cout, 3, 3)
    wsf = torch.empty(nf // 4 + 64, device=dev)
    fnew = lambda: ops.call("tsii_head_cat_fwd_low", ptr(low), ptr(skip), c1, c2, ptr(r0l), None, ptr(wt), ptr(bias), ptr(denom), ptr(keep), n, h, w, cout, ptr(yo), st)
    fold = lambda: ops.call("tsii_head_cat_fwd", ptr(low), ptr(skip), c1, c2, ptr(r0), None, ptr(wt), ptr(bias), ptr(denom), ptr(keep), n, h, w, cout, ptr(yo), ptr(wsf), nf, st)
    fnew(); ya = yo.clone(); fold(); yb = yo.clone()
    print(f"lib={os.environ.get('TSII_LIBRARY', 'default')}  forward matrix-core {timed(fnew):.3f} ms   vector-ALU {timed(fold):.3f} ms   max rel diff {float((ya - yb).abs().max() / yb.abs().max()):.2e}")
    new(); a = dw.clone(); old(); b = dw.clone()
    print(f"lib={os.environ.get('TSII_LIBRARY', 'default')}  dW matrix-core {timed(new):.3f} ms   vector-ALU {timed(old):.3f} ms   (no dbias; incl. the partial-row reduction)   max rel diff {float((a - b).abs().max() / b.abs().max()):.2e}")

if __name__ == "__main__":
    main()

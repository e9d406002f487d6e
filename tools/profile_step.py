#!/usr/bin/env python3
"""Per-entry-point / per-shape timing of one ImageFill training step (HIP events around every
C-ABI call on the launch stream).  Prints a table sorted by time; for the GEMM entry points also TF/s.

    python tools/profile_step.py [--batch 32] [--size 512]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--model", default="ImageFill")
    ap.add_argument("--pixel-shuffle", action="store_true")
    ap.add_argument("--storage", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--rows", type=int, default=70)
    ap.add_argument("--forward", action="store_true", help="time the forward pass only (train-mode BatchNorm + L1 loss under no_grad: bench.py's forward_only leg)")
    args = ap.parse_args()
    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.synthetic import make_batch
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    seg = args.model in ("TextSegament", "XceptionTextSegment")
    if seg:
        from text_segmentation_image_inpainting_amd.synthetic import make_seg_batch
        T.set_activation_storage(args.storage)
        net = getattr(T, args.model)(**({"pixel_shuffle_head": True} if (args.pixel_shuffle and args.model == "TextSegament") else {}))

        class SegStep(torch.nn.Module):
            def __init__(self, net):
                super().__init__()
                self.net = net

            def forward(self, a):
                return self.net(a[0])
        model = SegStep(net).to(dev).train()
        focal = T.BinaryFocalLoss(0, 1, 2)
        tr = FlatSGDTrainer(model, lr=1e-3, loss_fn=lambda out, tgt: focal(out, tgt))
        c, cl = (v.to(dev) for v in make_seg_batch(args.batch, args.size, seed0=0))
        m = None
    else:
        model = getattr(T, args.model)().to(dev).train()
        tr = FlatSGDTrainer(model, lr=1e-3)
        c, m, cl = make_batch(args.batch, args.size)
        c, m, cl = c.to(dev), m.to(dev), to_nhwc(cl.to(dev))
    for _ in range(2):
        tr.step(c, m, cl)
    torch.cuda.synchronize()
    _lib.start_timing(list(_lib.SIGNATURES))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    if args.forward:
        with torch.no_grad():
            tr.loss_fn(model((c, m)), cl)
    else:
        tr.step(c, m, cl)
    ev1.record()
    rec = _lib.stop_timing()
    total = ev0.elapsed_time(ev1)
    agg = collections.OrderedDict()
    for name, lst in rec.items():
        for ms, a in lst:
            key = (name, a)
            d = agg.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += ms
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    by_name = collections.Counter()
    for (name, a), (cnt, ms) in rows:
        by_name[name] += ms
    print(f"{'forward' if args.forward else 'step'} total {total:.2f} ms; sum of timed calls {sum(by_name.values()):.2f} ms")
    for name, ms in by_name.most_common():
        print(f"  {name:28s} {ms:8.2f} ms")
    print("--- per shape ---")
    for (name, a), (cnt, ms) in rows[:(200 if args.forward else args.rows)]:
        extra = ""
        if name in ("tsii_pw_fwd", "tsii_pw_bwd_dx", "tsii_pw_bwd_dw"):
            mm, p, q = a[0], a[1], a[2]
            extra = f"{2.0 * mm * p * q * cnt / (ms * 1e-3) / 1e12:6.1f} TF/s"
        print(f"{name:24s} x{cnt:<3d} {ms:8.3f} ms  {extra:>12s}  args={a}")


if __name__ == "__main__":
    main()

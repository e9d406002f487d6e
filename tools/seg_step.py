#!/usr/bin/env python3
"""Quick timing of a segmentation training step (TextSegament / XceptionTextSegment, BinaryFocalLoss).
    python tools/seg_step.py --model TextSegament --batch 8 --size 512
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="TextSegament")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--pixel-shuffle", action="store_true", help="TextSegament with the Conv(128,16)+PixelShuffle(4) head (cfg 3, SURVEY.md F3)")
    ap.add_argument("--checkpoint", action="store_true", help="recompute the encoder stages in backward (memory saver)")
    ap.add_argument("--products", type=int, default=-1, help="tsii_set_gemm_products (default: library default)")
    args = ap.parse_args()
    import text_segmentation_image_inpainting_amd as T
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from text_segmentation_image_inpainting_amd import _lib
    if args.products >= 0:
        _lib.set_gemm_products(args.products)
    kw = {"pixel_shuffle_head": True} if (args.pixel_shuffle and args.model == "TextSegament") else {}
    m = getattr(T, args.model)(**kw).to(dev).train()
    if args.checkpoint and hasattr(m, "checkpoint_encoder"):
        m.checkpoint_encoder = True
    x = torch.randn(args.batch, 3, args.size, args.size, device=dev)
    t = (torch.rand(args.batch, 1, args.size, args.size, device=dev) > 0.9).float()
    lossf = T.BinaryFocalLoss(0, 1, 2)
    for i in range(args.steps + 1):
        if i == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for p in m.parameters():
            p.grad = None
        loss = lossf(m(x), t)
        loss.backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"{args.model}{' +pixel-shuffle head' if kw else ''}{' +checkpointed encoder' if args.checkpoint else ''} products={_lib.get_gemm_products()} "
          f"{args.size}x{args.size} bs{args.batch}: {dt * 1e3:.1f} ms/step, {args.batch / dt:.1f} img/s, "
          f"loss {loss.item():.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()

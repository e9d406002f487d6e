#!/usr/bin/env python3
"""Host-pipeline throughput of the inpainting dataset (reference Dataloader.py:77-162), and one FED training
run: DataLoader workers -> pinned batches -> DevicePrefetcher -> the same step bench.py times with resident inputs.

  * img/s of ``ImageInpaintingData.__getitem__`` in one process (PIL decode + RandomResizedCrop with bicubic resize to 512^2 +
    threshold + 10x10 dilation + ``random_masks``), on synthetic PNG pages written to a temporary folder;
  * img/s of a ``DataLoader`` at several worker counts -> workers needed for one GPU (the bench's ~450 img/s) and for eight;
  * (with a GPU) img/s of the training loop fed that way, next to the resident-input rate.

    python tools/host_pipeline.py [--pages 64] [--workers 1,8,16,32] [--steps 12] [--no-gpu]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402


def write_pages(folder, pages, h=1170, w=827, seed=0):
    """Synthetic 'manga pages': grey panels + dark strokes (clean) and the text-difference image (mask) next to them."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(folder, "clean"), exist_ok=True)
    os.makedirs(os.path.join(folder, "mask"), exist_ok=True)
    for i in range(pages):
        page = np.full((h, w, 3), 235, np.uint8)
        for _ in range(12):
            y0, x0 = int(rng.integers(0, h - 100)), int(rng.integers(0, w - 100))
            page[y0:y0 + int(rng.integers(40, 300)), x0:x0 + int(rng.integers(40, 300))] = rng.integers(40, 220, size=3, dtype=np.uint8)
        page = np.clip(page.astype(np.int16) + rng.integers(-12, 12, size=page.shape, dtype=np.int16), 0, 255).astype(np.uint8)
        diff = np.zeros((h, w), np.uint8)
        for _ in range(25):
            y0, x0 = int(rng.integers(0, h - 40)), int(rng.integers(0, w - 120))
            diff[y0:y0 + int(rng.integers(8, 30)), x0:x0 + int(rng.integers(30, 110))] = 200
        Image.fromarray(page).save(os.path.join(folder, "clean", f"p{i:04d}.png"))
        Image.fromarray(diff).save(os.path.join(folder, "mask", f"p{i:04d}.png"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=64)
    ap.add_argument("--workers", default="1,8,16,32")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args()
    from text_segmentation_image_inpainting_amd.Dataloader import DevicePrefetcher, ImageInpaintingData, expand_compact_batch
    out = {"host_cores_logical": os.cpu_count(), "batch": args.batch}
    with tempfile.TemporaryDirectory() as tmp:
        write_pages(tmp, args.pages)
        ds = ImageInpaintingData(tmp, max_images=args.batch * 64, image_size=(512, 512), add_random_masks=True)
        torch.set_num_threads(1)
        ds[0]
        t0 = time.perf_counter()
        n = 48
        for i in range(n):
            ds[i]
        per_proc = n / (time.perf_counter() - t0)
        out["getitem_imgs_per_s_one_process"] = round(per_proc, 1)
        rates = {}
        for w in [int(v) for v in args.workers.split(",")]:
            if w > (os.cpu_count() or 1):
                continue
            dl = torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True, num_workers=w, pin_memory=False, drop_last=True,
                                             persistent_workers=False, prefetch_factor=4 if w else None)
            it = iter(dl)
            next(it)                                       # workers up, first batches queued
            t0 = time.perf_counter()
            k = 0
            for _ in range(max(4, min(3 * w, 40))):
                next(it)
                k += 1
            rates[w] = round(k * args.batch / (time.perf_counter() - t0), 1)
            del it, dl
        out["dataloader_imgs_per_s_by_workers"] = rates
        # per-worker rate where the loader still scales (beyond that one loader's main process -- collating 9 MB per image --
        # is the limit; under data parallelism every rank runs its OWN loader, so the per-rank requirement is what counts)
        per_worker = max(r / w for w, r in rates.items() if w >= 1)
        out["imgs_per_s_per_worker"] = round(per_worker, 1)
        out["workers_per_rank_for_450_imgs_per_s"] = int(np.ceil(1.15 * 450 / per_worker))
        out["workers_on_the_node_for_8_ranks"] = 8 * out["workers_per_rank_for_450_imgs_per_s"]
        if torch.cuda.is_available() and not args.no_gpu:
            import text_segmentation_image_inpainting_amd as T
            from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
            from text_segmentation_image_inpainting_amd.synthetic import make_batch
            from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
            dev = torch.device("cuda:0")
            torch.manual_seed(0)
            tr = FlatSGDTrainer(T.ImageFill().to(dev).train(), lr=1e-3)
            c, m, cl = make_batch(args.batch, 512)
            c, m, cl = c.to(dev), m.to(dev), to_nhwc(cl.to(dev))
            for _ in range(3):
                tr.step(c, m, cl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tr.step(c, m, cl)
            torch.cuda.synchronize()
            out["resident_imgs_per_s"] = round(args.steps * args.batch / (time.perf_counter() - t0), 1)
            w = min(max(rates), int(np.ceil(2.0 * out["resident_imgs_per_s"] / per_worker)))
            dl = torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True, num_workers=w, pin_memory=True, drop_last=True,
                                             prefetch_factor=4)
            feed = iter(DevicePrefetcher(dl, dev))
            for _ in range(3):
                cc, mm, cll = next(feed)
                tr.step(cc, mm, to_nhwc(cll))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cc, mm, cll = next(feed)
                tr.step(cc, mm, to_nhwc(cll))
            torch.cuda.synchronize()
            out["fed_imgs_per_s"] = round(args.steps * args.batch / (time.perf_counter() - t0), 1)
            out["fed_workers"] = w
            del feed, dl
            # the same loop fed with compact items (uint8 image + 1-channel uint8 mask, expanded on the GPU: a ninth of the bytes)
            dsc = ImageInpaintingData(tmp, max_images=args.batch * 64, image_size=(512, 512), add_random_masks=True, compact=True)
            dl = torch.utils.data.DataLoader(dsc, batch_size=args.batch, shuffle=True, num_workers=w, pin_memory=True, drop_last=True,
                                             prefetch_factor=4)
            feed = iter(DevicePrefetcher(dl, dev, expand=expand_compact_batch))
            for _ in range(3):
                cc, mm, cll = next(feed)
                tr.step(cc, mm, to_nhwc(cll))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2 * args.steps):
                cc, mm, cll = next(feed)
                tr.step(cc, mm, to_nhwc(cll))
            torch.cuda.synchronize()
            out["fed_compact_imgs_per_s"] = round(2 * args.steps * args.batch / (time.perf_counter() - t0), 1)
            del feed, dl
    print(json.dumps(out))


if __name__ == "__main__":
    main()

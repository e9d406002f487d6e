#!/usr/bin/env python3
"""Inference demo with the reference's flow (Examples/demo_segmentation.py:28-70): EvaluateSet -> model ->
sigmoid > 0.5 -> 3x3 max-pool -> un-pad / resize to the original size -> save mask and a convex-hull overlay.
PIL + scipy replace cv2 / torchvision.  Needs an MI355X (the models have no CPU path).

    python examples/demo_segmentation.py --img-folder test_data [--checkpoint ckpt.pt] [--model XceptionTextSegment]
    python examples/demo_segmentation.py --synthetic            # seeded manga-like tile, random-init weights
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image, ImageDraw  # noqa: E402


def max_pool3x3_binary(mask: torch.Tensor) -> torch.Tensor:
    """nn.MaxPool2d(3, stride 1, padding 1) on a {0,1} mask [N,1,H,W] (demo_segmentation.py:35)."""
    m = torch.nn.functional.pad(mask.float(), (1, 1, 1, 1), value=0)
    out = torch.zeros_like(mask, dtype=torch.float32)
    for dy in range(3):
        for dx in range(3):
            out = torch.maximum(out, m[..., dy:dy + mask.shape[-2], dx:dx + mask.shape[-1]])
    return out


def draw_bounding_box(img_np, mask_np, area_threshold=100):
    """Fill the convex hull of every connected mask region larger than ``area_threshold`` (demo_segmentation.py:17-25)."""
    from scipy import ndimage
    from scipy.spatial import ConvexHull
    lab, n = ndimage.label(mask_np > 127)
    canvas = Image.fromarray(img_np)
    draw = ImageDraw.Draw(canvas)
    for k in range(1, n + 1):
        ys, xs = np.nonzero(lab == k)
        if len(ys) <= area_threshold:
            continue
        pts = np.stack([xs, ys], 1).astype(np.float64)
        try:
            hull = ConvexHull(pts)
            poly = [tuple(pts[v]) for v in hull.vertices]
        except Exception:  # noqa: BLE001 - degenerate (collinear) region
            poly = [(xs.min(), ys.min()), (xs.max(), ys.min()), (xs.max(), ys.max()), (xs.min(), ys.max())]
        draw.polygon(poly, fill=(50, 128, 30))
    return np.asarray(canvas)


def process(model, eval_img, device):
    (img, origin, unpadder), file_name = eval_img
    with torch.no_grad():
        out = model(img.to(device))
    mask = (out > 0).cpu()                       # sigmoid(out) > 0.5
    mask = max_pool3x3_binary(mask).byte()
    mask = unpadder(mask).float()
    Image.fromarray((mask[0, 0].numpy() * 255).astype(np.uint8)).save(file_name + "_mask.jpg")
    origin_np = (origin[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    mask_np = (mask[0, 0].numpy() * 255).astype(np.uint8)
    Image.fromarray(draw_bounding_box(origin_np, mask_np, 500)).save(file_name + "_contour.jpg")
    return mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="XceptionTextSegment", choices=["XceptionTextSegment", "TextSegament"])
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--img-folder", default=None)
    ap.add_argument("--resize", type=int, default=600)
    ap.add_argument("--synthetic", action="store_true")
    args = ap.parse_args()
    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd.Dataloader import EvaluateSet
    from text_segmentation_image_inpainting_amd.synthetic import manga_tile
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = getattr(T, args.model)()
    model.total_parameters()
    if args.checkpoint:
        # checkpoints trained with in-place ABN are compatible with plain BatchNorm keys (demo_segmentation.py:55)
        model.load_state_dict(torch.load(args.checkpoint, map_location="cpu"))
    model = model.to(dev).eval()
    folder = args.img_folder
    if args.synthetic or folder is None:
        folder = tempfile.mkdtemp(prefix="tsii_demo_")
        tile = (manga_tile(256, np.random.default_rng(0)).transpose(1, 2, 0) * 255).astype(np.uint8)
        Image.fromarray(tile).save(os.path.join(folder, "tile.png"))
    evalset = EvaluateSet(mean=[0.4935, 0.4563, 0.4544], std=[0.3769, 0.3615, 0.3566], img_folder=folder, resize=args.resize)
    t0 = time.time()
    for item in evalset:
        m = process(model, item, dev)
        print(item[1], "mask", tuple(m.shape), "text fraction %.4f" % float(m.float().mean()))
    torch.cuda.synchronize()
    print("Runtime :{}".format(time.time() - t0))


if __name__ == "__main__":
    main()

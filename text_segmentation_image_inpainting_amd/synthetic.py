"""Seeded synthetic inpainting samples with the reference Dataset's tuple contract
(``ImageInpaintingData.__getitem__`` -> ``(corrupted, binary_mask, clean)``, each float32
[3, S, S], mask in {0,1} with 1 = valid; Dataloader.py:103-132) -- PIL + numpy only (cv2 and
torchvision are not available).

The deterministic part of the reference pipeline is restated exactly: threshold at
``0.4 * 255`` (:30,120), 10x10 dilation with the anchor at (5,5) i.e. max over offsets -5..+4
(:121), ``1 - mask`` expanded to 3 channels (:128-129), ``corrupted = clean * mask`` (:131).
The random part (``random_masks``: 1-5 lines of width 15-20 whose second point lies within
+-75 px, 1-5 ellipses of 20-70 px, :142-162) is matched in distribution, not bit pattern
(its RNG streams live in un-pinned third-party code).
"""
import numpy as np
import torch
from PIL import Image, ImageDraw

BRIGHTNESS_DIFFERENCE = 0.4  # Dataloader.py:30


def random_masks(size=512, offset=10, rng=None) -> Image.Image:
    rng = rng or np.random.default_rng()
    img = Image.new("L", (size, size), 0)
    draw = ImageDraw.Draw(img)
    for _ in range(int(rng.integers(1, 6))):
        p0 = rng.integers(offset, size, size=2)
        p1 = np.clip(rng.integers(offset, size, size=2), p0 - 75, p0 + 75)
        draw.line([int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])], width=int(rng.integers(15, 21)), fill=255)
    for _ in range(int(rng.integers(1, 6))):
        c = np.sort(rng.integers(offset, size - offset, size=2))
        e = np.clip(rng.integers(20, 70, size=2) + c, offset, size - offset)
        box = [int(min(c[0], e[0])), int(min(c[1], e[1])), int(max(c[0], e[0])), int(max(c[1], e[1]))]
        draw.ellipse(box, fill=255)
    return img


def dilate_10x10(mask_u8: np.ndarray) -> np.ndarray:
    """cv2.dilate(mask, ones((10,10))) : max over row/col offsets -5..+4 (anchor (5,5)), zero border."""
    h, w = mask_u8.shape
    pad = np.zeros((h + 9, w + 9), dtype=mask_u8.dtype)
    pad[5:5 + h, 5:5 + w] = mask_u8
    out = np.zeros_like(mask_u8)
    rows = np.zeros((h + 9, w), dtype=mask_u8.dtype)
    for dx in range(10):
        np.maximum(rows, pad[:, dx:dx + w], out=rows)
    for dy in range(10):
        np.maximum(out, rows[dy:dy + h], out=out)
    return out


def manga_tile(size=512, rng=None) -> np.ndarray:
    """Grey 'paper' in [0.7, 1.0] with 20-40 dark strokes / boxes; float32 [3, S, S] in [0, 1];
    40 % of the tiles are grey-scale (RandomGrayscale(p=0.4), Dataloader.py:93)."""
    rng = rng or np.random.default_rng()
    base = rng.uniform(0.7, 1.0)
    chans = []
    grey = rng.uniform() < 0.4
    img = Image.new("RGB", (size, size), tuple(int(255 * base) for _ in range(3)))
    draw = ImageDraw.Draw(img)
    for _ in range(int(rng.integers(20, 41))):
        x0, y0 = (int(v) for v in rng.integers(0, size, size=2))
        x1, y1 = (int(np.clip(v, 0, size - 1)) for v in (x0 + rng.integers(-120, 121), y0 + rng.integers(-120, 121)))
        shade = int(rng.integers(0, 90))
        col = (shade,) * 3 if grey else tuple(int(np.clip(shade + rng.integers(-30, 31), 0, 255)) for _ in range(3))
        if rng.uniform() < 0.7:
            draw.line([x0, y0, x1, y1], width=int(rng.integers(1, 7)), fill=col)
        else:
            draw.rectangle([min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1)], outline=col, width=int(rng.integers(1, 4)))
    arr = np.asarray(img, dtype=np.float32) / 255.0
    arr = arr + rng.normal(0, 0.01, arr.shape).astype(np.float32)
    return np.clip(arr, 0.0, 1.0).transpose(2, 0, 1).astype(np.float32)


def make_sample(seed: int, size=512, bernoulli=False):
    """(corrupted, binary_mask, clean) float32 [3,S,S] like the reference Dataset.  ``bernoulli``
    gives the stress variant of the reference's commented snippet (image_inpainting.py:95-97):
    i.i.d. per-pixel, per-channel masks."""
    rng = np.random.default_rng(seed)
    clean = manga_tile(size, rng)
    if bernoulli:
        binary = (rng.standard_normal((3, size, size)) > 0).astype(np.float32)
    else:
        m = np.asarray(random_masks(size, 10, rng), dtype=np.uint8)
        m = np.where(m > BRIGHTNESS_DIFFERENCE * 255, np.uint8(255), np.uint8(0))
        m = dilate_10x10(m)
        binary = 1.0 - (m.astype(np.float32) / 255.0)
        binary = np.broadcast_to(binary[None], (3, size, size)).copy()
    corrupted = clean * binary
    return corrupted, binary, clean


def make_batch(batch: int, size=512, seed0=0, bernoulli=False):
    """Collated batch (three float32 tensors [B,3,S,S]) -- what a DataLoader over the Dataset yields."""
    parts = [make_sample(seed0 + i, size, bernoulli) for i in range(batch)]
    return tuple(torch.from_numpy(np.stack([p[j] for p in parts])) for j in range(3))


SEG_MEAN = (0.4935, 0.4563, 0.4544)   # Examples/demo_segmentation.py:59-60
SEG_STD = (0.3769, 0.3615, 0.3566)


def make_seg_sample(seed: int, size=512):
    """(image, target) for the text-segmentation nets (SURVEY.md 8(d) cfg 1 / 3): a normalised manga tile
    float32 [3,S,S] and the binary 'text' mask float32 [1,S,S] = its dark strokes dilated by 5 px."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(0.7, 1.0)
    img = Image.new("L", (size, size), int(255 * base))
    tgt = Image.new("L", (size, size), 0)
    di, dt = ImageDraw.Draw(img), ImageDraw.Draw(tgt)
    for _ in range(int(rng.integers(20, 41))):
        x0, y0 = (int(v) for v in rng.integers(0, size, size=2))
        x1, y1 = (int(np.clip(v, 0, size - 1)) for v in (x0 + rng.integers(-120, 121), y0 + rng.integers(-120, 121)))
        wd = int(rng.integers(1, 7))
        di.line([x0, y0, x1, y1], width=wd, fill=int(rng.integers(0, 90)))
        dt.line([x0, y0, x1, y1], width=wd + 10, fill=255)
    arr = np.asarray(img, dtype=np.float32) / 255.0
    arr = np.clip(arr + rng.normal(0, 0.01, arr.shape).astype(np.float32), 0.0, 1.0)
    x = np.stack([(arr - m) / s for m, s in zip(SEG_MEAN, SEG_STD)]).astype(np.float32)
    t = (np.asarray(tgt, dtype=np.uint8) > 0).astype(np.float32)[None]
    return x, t


def make_seg_batch(batch: int, size=512, seed0=0):
    parts = [make_seg_sample(seed0 + i, size) for i in range(batch)]
    return tuple(torch.from_numpy(np.stack([p[j] for p in parts])) for j in range(2))

"""Host-side datasets with the reference's tuple contracts (Dataloader.py:41-162,265-317), using PIL + numpy +
torch only (cv2 / torchvision are not available).

Deterministic steps are restated exactly and pinned by known-answer tests (tests/test_dataloader.py):
threshold at ``brightness_difference * 255`` (:30,120), 10x10 dilation with anchor (5,5) (:121), ``1 - mask``
expanded to 3 channels (:128-129), ``corrupted = clean * mask`` (:131), ``EvaluateSet`` resize / floor-to-8 /
pad-right-or-bottom geometry (:285-305) and its un-pad + bilinear ``resize_mask`` closure (:307-317).
The random augmentations (RandomResizedCrop parameters, ColorJitter, RandomGrayscale, ``random_masks``) follow the
same distributions through Python's ``random`` (the reference's RNG streams live in un-pinned third-party code,
so bit patterns are not reproducible -- "parity unpinned").
"""
import glob
import math
import os
import random
import re
from itertools import chain

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageEnhance
from torch.nn.functional import interpolate, pad
from torch.utils.data import Dataset

from .synthetic import dilate_10x10

brightness_difference = 0.4  # in [0,1]  (Dataloader.py:30)


def to_tensor(pic) -> torch.Tensor:
    """PIL image / HxW(xC) uint8 array -> float32 [C,H,W] in [0,1] (torchvision ``to_tensor``)."""
    arr = np.array(pic)  # copy: PIL buffers are read-only
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


def random_resized_crop_params(img, scale, ratio):
    """torchvision ``RandomResizedCrop.get_params``: -> (i, j, h, w)."""
    width, height = img.size
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * random.uniform(scale[0], scale[1])
        aspect = math.exp(random.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return random.randint(0, height - h), random.randint(0, width - w), h, w
    in_ratio = float(width) / float(height)      # fallback: central crop
    if in_ratio < min(ratio):
        w, h = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        h, w = height, int(round(height * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def resized_crop(img, i, j, h, w, size, interpolation=Image.BICUBIC):
    return img.crop((j, i, j + w, i + h)).resize((size[1], size[0]), interpolation)


def color_jitter_L(img, brightness=0.2, contrast=0.2):
    """ColorJitter(0.2 x4) on an 'L' image: saturation / hue are no-ops for one channel; brightness and contrast
    factors are U(0.8, 1.2), applied in random order."""
    ops_ = [(ImageEnhance.Brightness, random.uniform(1 - brightness, 1 + brightness)),
            (ImageEnhance.Contrast, random.uniform(1 - contrast, 1 + contrast))]
    random.shuffle(ops_)
    for enh, f in ops_:
        img = enh(img).enhance(f)
    return img


def random_masks(pil_img, size=512, offset=10):
    """Irregular holes: 1-5 lines (width 15-20, second point within +-75 px) and 1-5 ellipses of 20-70 px
    (Dataloader.py:142-162)."""
    draw = ImageDraw.Draw(pil_img)
    for _ in range(random.randint(1, 5)):
        cords = np.array(random.choices(range(offset, size), k=4)).reshape(2, 2)
        cords[1] = np.clip(cords[1], a_min=cords[0] - 75, a_max=cords[0] + 75)
        draw.line(cords.reshape(-1).tolist(), width=random.randint(15, 20), fill=255)
    for _ in range(random.randint(1, 5)):
        cords = np.array(random.choices(range(offset, size - offset), k=2))
        cords.sort()
        ex = np.clip(np.array(random.choices(range(20, 70), k=2)) + cords, a_min=offset, a_max=size - offset)
        box = np.concatenate([cords, ex]).tolist()
        draw.ellipse([min(box[0], box[2]), min(box[1], box[3]), max(box[0], box[2]), max(box[1], box[3])], fill=255)
    return pil_img


def hole_map_from_difference(mask_pil):
    """uint8 'L' difference image -> uint8 [H,W] map, 255 = hole (threshold + 10x10 dilation, Dataloader.py:120-121)."""
    m = np.where(np.array(mask_pil) > brightness_difference * 255, np.uint8(255), np.uint8(0))
    return dilate_10x10(m)


def binary_mask_from_difference(mask_pil):
    """uint8 'L' difference image -> float32 [3,H,W] mask with 1 = valid, 0 = hole (Dataloader.py:120-129)."""
    mask_t = to_tensor(hole_map_from_difference(mask_pil)[:, :, None])
    return (1 - mask_t).expand(3, -1, -1)


def expand_compact_batch(clean_u8: torch.Tensor, valid_u8: torch.Tensor):
    """(clean [N,3,H,W] uint8, valid [N,1,H,W] uint8 in {0, 1}) -> the reference's float32 (corrupted, mask, clean) triple, on
    whatever device the inputs live on: ``clean = u8 / 255`` is torchvision's ``to_tensor`` (the same IEEE division), the mask is
    exactly 0 / 1 and ``corrupted = clean * mask`` (Dataloader.py:128-131) -- bit-identical to the float pipeline at a ninth of the
    bytes through worker shared memory, the pin-memory thread and PCIe (4 instead of 36 bytes per pixel)."""
    clean = clean_u8.float().div(255)
    mask = valid_u8.float().expand(-1, 3, -1, -1).contiguous()
    return clean * mask, mask, clean


class TextSegmentationData(Dataset):
    """(raw [1,H,W], mask [1,H,W]) float32 pairs (Dataloader.py:41-74); masks live in the sibling 'mask' folder."""

    def __init__(self, img_raw_folder, image_size=(256, 256)):
        super().__init__()
        self.raw_images = glob.glob(os.path.join(img_raw_folder, "*"))
        assert len(self.raw_images) > 0
        print("Find {} images. ".format(len(self.raw_images)))
        self.img_size = image_size

    def __len__(self):
        return len(self.raw_images)

    def __getitem__(self, item):
        img_file = self.raw_images[item]
        img_raw = Image.open(img_file).convert("L")
        img_mask = Image.open(re.sub("raw", "mask", img_file)).convert("L")
        return self.process_images(img_raw, img_mask)

    def process_images(self, raw, clean):
        i, j, h, w = random_resized_crop_params(raw, scale=(0.1, 2), ratio=(3. / 4., 4. / 3.))
        raw_img = color_jitter_L(resized_crop(raw, i, j, h, w, self.img_size))
        mask_img = resized_crop(clean, i, j, h, w, self.img_size)
        return to_tensor(raw_img), to_tensor(mask_img)


class ImageInpaintingData(Dataset):
    """(corrupted, binary_mask, clean), each float32 [3,H,W]; mask 1 = valid (Dataloader.py:77-139)."""

    def __init__(self, image_folder, max_images=False, image_size=(512, 512), add_random_masks=False, compact=False):
        """``compact`` (not in the reference): items are (clean [3,H,W] uint8, valid [1,H,W] uint8) instead of the three float32
        tensors; ``expand_compact_batch`` (e.g. as ``DevicePrefetcher(..., expand=expand_compact_batch)``) rebuilds the reference's
        triple bit for bit on the GPU.  At ~480 img/s per GPU the float path's 288 MB per batch of 32 is what limits a
        ``DataLoader`` process (tools/host_pipeline.py)."""
        super().__init__()
        self.compact = bool(compact)
        if isinstance(image_folder, str):
            self.images = glob.glob(os.path.join(image_folder, "clean/*"))
        else:
            self.images = list(chain.from_iterable([glob.glob(os.path.join(i, "clean/*")) for i in image_folder]))
        assert len(self.images) > 0
        if max_images:
            self.images = random.choices(self.images, k=max_images)
        print(f"Find {len(self.images)} images.")
        self.img_size = image_size
        self.add_random_masks = add_random_masks

    def __len__(self):
        return len(self.images)

    def __getitem__(self, item):
        img_file = self.images[item]
        img_clean = Image.open(img_file).convert("RGB")
        img_mask = Image.open(re.sub("clean", "mask", img_file)).convert("L")
        return self.process_images(img_clean, img_mask)

    def process_images(self, clean, mask):
        i, j, h, w = random_resized_crop_params(clean, scale=(0.5, 2.0), ratio=(3. / 4., 4. / 3.))
        clean_img = resized_crop(clean, i, j, h, w, self.img_size)
        mask = resized_crop(mask, i, j, h, w, self.img_size)
        if self.add_random_masks:
            mask = random_masks(mask.copy(), size=self.img_size[0], offset=10)
        holes = hole_map_from_difference(mask)
        if random.random() < 0.4:                                   # RandomGrayscale(p=0.4)
            clean_img = clean_img.convert("L").convert("RGB")
        if self.compact:
            clean_u8 = torch.from_numpy(np.ascontiguousarray(np.array(clean_img).transpose(2, 0, 1)))
            return clean_u8, torch.from_numpy((holes == 0).astype(np.uint8))[None]
        binary_mask = (1 - to_tensor(holes[:, :, None])).expand(3, -1, -1)
        clean_t = to_tensor(clean_img)
        return clean_t * binary_mask, binary_mask, clean_t

    @staticmethod
    def get_mask(raw_pil, clean_pil):
        from PIL import ImageChops
        return ImageChops.difference(raw_pil.convert("L"), clean_pil.convert("L"))


class EvaluateSet(Dataset):
    """Inference set (Dataloader.py:265-317): ((img [1,3,R,R-ish], origin [1,3,H,W], mask_resizer), path)."""

    def __init__(self, mean, std, img_folder=None, resize=512):
        self.eval_imgs = [glob.glob(img_folder + "/*.{}".format(i), recursive=True) for i in ["jpg", "jpeg", "png"]]
        self.eval_imgs = list(chain.from_iterable(self.eval_imgs))
        assert resize % 8 == 0
        self.resize = resize
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        print("Find {} test images. ".format(len(self.eval_imgs)))

    def __len__(self):
        return len(self.eval_imgs)

    def __getitem__(self, item):
        img_file = self.eval_imgs[item]
        img = Image.open(img_file).convert("RGB")
        return self.resize_pad_tensor(img), img_file

    def resize_pad_tensor(self, pil_img):
        origin = to_tensor(pil_img).unsqueeze(0)
        fix_len = self.resize
        ratio = fix_len / max(pil_img.size)
        new_size = tuple(map(lambda x: int(x * ratio) // 8 * 8, pil_img.size))
        img = pil_img.resize(new_size, Image.BICUBIC)
        img = ((to_tensor(img) - self.mean) / self.std).unsqueeze(0)
        _, _, h, w = img.size()
        boarder_pad = (0, fix_len - w, 0, 0) if fix_len > w else (0, 0, 0, fix_len - h)
        img = pad(img, boarder_pad, value=0)
        return img, origin, self.resize_mask(boarder_pad, pil_img.size)

    @staticmethod
    def resize_mask(padded_values, origin_size):
        """closure: padded-size mask -> un-pad -> bilinear to the original size -> (> 0) expanded to 3 channels."""
        left, right, top, bottom = padded_values

        def m(x):
            x = x[..., top: x.shape[-2] - bottom if bottom else None, left: x.shape[-1] - right if right else None]
            x = interpolate(x.float(), size=tuple(reversed(origin_size)), mode="bilinear", align_corners=False)
            return x.expand(-1, 3, -1, -1) > 0
        return m


class DevicePrefetcher:
    """Feeds a training loop from a ``DataLoader`` of (corrupted, mask, clean) tuples: batch k+1 is
    copied host -> HBM on a side HIP stream (from pinned memory, so the copy is asynchronous) while the kernels of batch k
    run on the compute stream.  At ~310 img/s the three 3 MB/img tensors are 2.8 GB/s -- 4 % of PCIe Gen5 x16 -- so the
    copy disappears behind the step.

        for corrupted, mask, clean in DevicePrefetcher(loader, device):
            trainer.step(corrupted, mask, to_nhwc(clean))
    """

    def __init__(self, loader, device, expand=None):
        """``expand``: applied to the uploaded batch on the copy stream (e.g. ``expand_compact_batch`` for a ``compact`` dataset)."""
        self.loader, self.device, self.expand = loader, torch.device(device), expand
        self.stream = torch.cuda.Stream(device=self.device)

    def _upload(self, batch):
        out = []
        with torch.cuda.stream(self.stream):
            for t in batch:
                if torch.is_tensor(t):
                    if not t.is_pinned():
                        t = t.pin_memory()
                    out.append(t.to(self.device, non_blocking=True))
                else:
                    out.append(t)
            if self.expand is not None:
                out = self.expand(*out)
        return tuple(out)

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        for batch in it:
            cur = nxt
            torch.cuda.current_stream(self.device).wait_stream(self.stream)   # batch k has landed
            for t in cur:
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream(self.device))
            nxt = self._upload(batch)                                          # batch k+1 flies during step k
            yield cur
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        for t in nxt:
            if torch.is_tensor(t):
                t.record_stream(torch.cuda.current_stream(self.device))
        yield nxt

"""Known-answer tests for the host-side datasets (a21 / a22): the deterministic geometry and mask pipeline of
Dataloader.py restated without cv2 / torchvision."""
import os

import pytest

import numpy as np
import torch
from PIL import Image

from text_segmentation_image_inpainting_amd import Dataloader as D


def _save(arr, path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def test_inpainting_dataset_contract(tmp_path):
    rng = np.random.default_rng(0)
    clean = rng.integers(0, 255, (96, 128, 3), dtype=np.uint8)
    diff = np.zeros((96, 128), np.uint8)
    diff[30:50, 40:70] = 200
    _save(clean, str(tmp_path / "clean" / "a.png"))
    _save(diff, str(tmp_path / "mask" / "a.png"))
    ds = D.ImageInpaintingData(str(tmp_path), image_size=(64, 64), add_random_masks=True)
    corrupted, mask, clean_t = ds[0]
    assert corrupted.shape == mask.shape == clean_t.shape == (3, 64, 64) and corrupted.dtype == torch.float32
    assert set(np.unique(mask.numpy())) <= {0.0, 1.0}
    assert torch.equal(corrupted, clean_t * mask)                                  # Dataloader.py:131
    assert torch.equal(mask[0], mask[1]) and torch.equal(mask[0], mask[2])         # expand(3,-1,-1) (:129)
    assert float(mask.min()) == 0.0                                                # random_masks always adds holes


def test_compact_items_expand_to_the_float_triple_bit_for_bit(tmp_path):
    """``compact=True`` (uint8 transport, not in the reference): with the same RNG stream the expanded batch equals the float
    pipeline's (corrupted, mask, clean) exactly -- /255 is the same division, the mask exactly 0 / 1."""
    import random
    rng = np.random.default_rng(2)
    for name in ("a", "b", "c"):
        clean = rng.integers(0, 255, (96, 128, 3), dtype=np.uint8)
        diff = np.zeros((96, 128), np.uint8)
        diff[20:60, 30:90] = rng.integers(0, 255, (40, 60), dtype=np.uint8)
        _save(clean, str(tmp_path / "clean" / f"{name}.png"))
        _save(diff, str(tmp_path / "mask" / f"{name}.png"))
    ds_f = D.ImageInpaintingData(str(tmp_path), image_size=(64, 64), add_random_masks=True)
    ds_c = D.ImageInpaintingData(str(tmp_path), image_size=(64, 64), add_random_masks=True, compact=True)
    ds_c.images = list(ds_f.images)
    for i in range(6):
        random.seed(100 + i)
        corrupted, mask, clean_t = ds_f[i % 3]
        random.seed(100 + i)
        cu8, vu8 = ds_c[i % 3]
        assert cu8.dtype == vu8.dtype == torch.uint8 and cu8.shape == (3, 64, 64) and vu8.shape == (1, 64, 64)
        c2, m2, k2 = D.expand_compact_batch(cu8[None], vu8[None])
        assert torch.equal(k2[0], clean_t) and torch.equal(m2[0], mask) and torch.equal(c2[0], corrupted)
        assert c2.dtype == torch.float32 and m2.is_contiguous()


def test_binary_mask_threshold_and_dilation():
    m = np.zeros((40, 40), np.uint8)
    m[20, 20] = 103          # > 0.4*255 = 102  -> hole seed
    m[5, 5] = 102            # not above the threshold
    b = D.binary_mask_from_difference(Image.fromarray(m))
    hole = (b[0] == 0).numpy()
    ys, xs = np.nonzero(hole)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (16, 25, 16, 25) and hole.sum() == 100   # 10x10, anchor (5,5)
    assert b.shape == (3, 40, 40)


def test_segmentation_dataset_contract(tmp_path):
    rng = np.random.default_rng(1)
    _save(rng.integers(0, 255, (80, 100), dtype=np.uint8), str(tmp_path / "raw" / "p.png"))
    _save(rng.integers(0, 255, (80, 100), dtype=np.uint8), str(tmp_path / "mask" / "p.png"))
    ds = D.TextSegmentationData(str(tmp_path / "raw"), image_size=(32, 48))
    raw, mask = ds[0]
    assert raw.shape == mask.shape == (1, 32, 48) and 0.0 <= float(raw.min()) and float(raw.max()) <= 1.0


def test_evaluate_set_geometry(tmp_path):
    _save(np.full((300, 500, 3), 128, np.uint8), str(tmp_path / "wide.png"))
    _save(np.full((500, 300, 3), 128, np.uint8), str(tmp_path / "tall.png"))
    ev = D.EvaluateSet(mean=[0.4935, 0.4563, 0.4544], std=[0.3769, 0.3615, 0.3566], img_folder=str(tmp_path), resize=600)
    got = {os.path.basename(p): item for item, p in (ev[i] for i in range(len(ev)))}
    img, origin, unpad = got["wide.png"]            # 500x300 -> ratio 1.2 -> (600, 360) -> floor8 (600, 360); pad bottom
    assert img.shape == (1, 3, 600, 600) and origin.shape == (1, 3, 300, 500)
    assert torch.all(img[:, :, 360:, :] == 0) and not torch.all(img[:, :, :360, :] == 0)
    m = unpad(torch.ones(1, 1, 600, 600))
    assert m.shape == (1, 3, 300, 500) and m.dtype == torch.bool and bool(m.all())
    img, origin, unpad = got["tall.png"]            # 300x500 -> (360, 600); pad right
    assert img.shape == (1, 3, 600, 600) and torch.all(img[:, :, :, 360:] == 0)
    half = torch.zeros(1, 1, 600, 600)
    half[..., :180] = 1                              # left half of the un-padded 360 columns
    m = unpad(half)
    assert m.shape == (1, 3, 500, 300)
    assert bool(m[0, 0, :, :140].all()) and not bool(m[0, 0, :, 160:].any())
    # normalisation known answer: (128/255 - mean) / std
    exp = (128 / 255 - 0.4935) / 0.3769
    assert abs(float(got["wide.png"][0][0, 0, 10, 10]) - exp) < 1e-5


def test_demo_postprocessing_known_answers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("demo", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "demo_segmentation.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    m = torch.zeros(1, 1, 8, 8)
    m[0, 0, 3, 4] = 1
    d = demo.max_pool3x3_binary(m)
    assert torch.equal(d, torch.nn.functional.max_pool2d(m, 3, 1, 1)) and float(d.sum()) == 9.0
    img = np.zeros((40, 40, 3), np.uint8)
    mask = np.zeros((40, 40), np.uint8)
    mask[10:20, 10:30] = 255
    out = demo.draw_bounding_box(img, mask, area_threshold=50)
    assert tuple(out[15, 20]) == (50, 128, 30) and tuple(out[30, 5]) == (0, 0, 0)


@pytest.mark.gpu
def test_device_prefetcher_gpu():
    """n1: side-stream H2D prefetch delivers exactly the loader's batches, in order, on the device."""
    import torch
    from text_segmentation_image_inpainting_amd.Dataloader import DevicePrefetcher
    batches = [(torch.full((2, 3, 8, 8), float(i)), torch.ones(2, 3, 8, 8) * (i % 2), torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).reshape(2, 3, 8, 8) + i)
               for i in range(5)]
    got = list(DevicePrefetcher(batches, "cuda:0"))
    assert len(got) == 5
    for i, (a, b, c) in enumerate(got):
        assert a.is_cuda and b.is_cuda and c.is_cuda
        torch.cuda.synchronize()
        assert torch.equal(a.cpu(), batches[i][0]) and torch.equal(b.cpu(), batches[i][1]) and torch.equal(c.cpu(), batches[i][2])
    assert list(DevicePrefetcher([], "cuda:0")) == []

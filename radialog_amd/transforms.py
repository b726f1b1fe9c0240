"""Inference image transform of the hot path, host side, PIL + numpy only.

Reference: `create_chest_xray_transform_for_inference(512, center_crop_size=448)` = Compose([Resize(512), CenterCrop(448),
ToTensor(), ExpandChannels()]) (model/lavis/data/ReportDataset.py:80-106, used at demo.py:144,:251; the findings classifier
uses center_crop_size=488, demo.py:169), applied to the PIL "L" image that `load_image` / `remap_to_uint8` produce
(demo.py:173-218). Resize / CenterCrop / ToTensor are torchvision==0.14.0 transforms (requirements.txt:18, not vendored,
absent here), restated from their published arithmetic:

  Resize(int s)      the SHORTER side becomes s, the longer one int(s * long / short) -- truncation, not rounding
                     (torchvision.transforms.functional._compute_resized_output_size); PIL bilinear resampling
  CenterCrop(c)      top = int(round((H - c) / 2.0)), left = int(round((W - c) / 2.0)) -- Python's round (half to even)
  ToTensor()         uint8 [H, W] -> float32 [1, H, W] / 255
  ExpandChannels()   repeat_interleave to 3 channels; anything but one input channel is a ValueError

Round 6: with an engine the transform runs on the GPU (rdx_transform_image, api_transform.hip) -- same bytes; oracle/pil_resize.py is the numpy restatement of
Pillow's algorithm both are tested against. Benchmarks feed synthetic 448 x 448 tensors and bypass this file."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch


def resized_size(w: int, h: int, size: int) -> Tuple[int, int]:
    """(new_w, new_h) of torchvision Resize(size) on a (w, h) image."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def center_crop_box(w: int, h: int, crop: int) -> Tuple[int, int, int, int]:
    """(left, top, right, bottom) of torchvision CenterCrop(crop) on a (w, h) image that is at least crop x crop."""
    if w < crop or h < crop:
        raise ValueError(f"image {w}x{h} smaller than the {crop} px crop (torchvision would zero-pad; no RaDialog input does)")
    top = int(round((h - crop) / 2.0))
    left = int(round((w - crop) / 2.0))
    return left, top, left + crop, top + crop


class ExpandChannels:
    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        if data.shape[0] != 1:
            raise ValueError(f"Expected input of shape [1, H, W], found {data.shape}")
        return torch.repeat_interleave(data, 3, dim=0)


class ChestXrayInferenceTransform:
    """Callable like the reference's Compose: PIL image (mode "L") -> float32 [3, crop, crop] in [0, 1].
    `engine` (an RdxEngine, at construction or per call): the transform runs on that engine's GPU (librdx rdx_transform_image: the two integer resampling passes
    of Pillow's Resample.c + crop + /255 as HIP kernels, bit for bit the host result) and the tensor comes back on the device; without one it is the host path
    below -- the library call the reference itself makes (torchvision's Resize on a PIL image is PIL.Image.resize)."""

    def __init__(self, resize: int, center_crop_size: int, engine=None):
        self.resize, self.crop, self.engine = int(resize), int(center_crop_size), engine

    def __call__(self, img, engine=None) -> torch.Tensor:
        from PIL import Image
        engine = engine or self.engine
        if engine is not None:
            arr = np.asarray(img)
            if arr.ndim != 2 or arr.dtype != np.uint8:
                raise ValueError(f"Expected input of shape [1, H, W], found {arr.shape[::-1]}")
            return engine.transform_image(torch.from_numpy(np.array(arr, dtype=np.uint8)), self.resize, self.crop)      # (a writable copy: PIL hands out a read-only buffer)
        w, h = img.size
        img = img.resize(resized_size(w, h, self.resize), Image.BILINEAR)
        img = img.crop(center_crop_box(*img.size, self.crop))
        arr = np.asarray(img, dtype=np.uint8)
        if arr.ndim != 2:
            raise ValueError(f"Expected input of shape [1, H, W], found {arr.shape[::-1]}")
        x = torch.from_numpy(arr.astype(np.float32) / 255.0)[None]
        return ExpandChannels()(x)


def create_chest_xray_transform_for_inference(resize: int, center_crop_size: int, engine=None) -> ChestXrayInferenceTransform:
    return ChestXrayInferenceTransform(resize, center_crop_size, engine)


def remap_to_uint8(array: np.ndarray) -> np.ndarray:
    """demo.py:173-202 with percentiles=None: min -> 0, max -> 255, truncating cast."""
    array = array.astype(float)
    array -= array.min()
    array /= array.max()
    array *= 255
    return array.astype(np.uint8)


def load_image(path):
    """demo.py:205-218: file -> remapped uint8 -> PIL "L" (the reference reads with skimage.io.imread, which for png / jpg is PIL)."""
    from PIL import Image
    return Image.fromarray(remap_to_uint8(np.asarray(Image.open(path)))).convert("L")

"""CPU ORACLE of the inference image transform -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Reference: `create_chest_xray_transform_for_inference(512, center_crop_size=448 | 488)` = Compose([Resize, CenterCrop, ToTensor, ExpandChannels])
(model/lavis/data/ReportDataset.py:80-106; call sites demo.py:144, :169, :251) on the PIL "L" image of demo.py:205-218. Resize / CenterCrop / ToTensor are
`torchvision==0.14.0` transforms (requirements.txt:18 -- third-party, absent here and un-vendored); on a PIL image torchvision's Resize IS
`PIL.Image.resize(size, BILINEAR)`, i.e. the arithmetic lives in Pillow's `src/libImaging/Resample.c` (the reference does not pin Pillow; the algorithm has been the
same since Pillow 4: this file is checked against the Pillow installed in the container, 12.2.0, bit for bit -- tests/test_transforms.py -- which pins it).
Restated from the published algorithm:

  Resize(s)        shorter side -> s, longer side -> int(s * long / short) (truncation; torchvision _compute_resized_output_size)
  PIL resize       two passes, horizontal then vertical, each a convolution with the BILINEAR (triangle) filter stretched by the down-scaling factor
                   (antialiasing: support = max(scale, 1)), coefficients computed in C doubles, normalised, converted to fixed point with PRECISION_BITS
                   = 32 - 8 - 2 = 22 ((int)(0.5 + k * 2^22)), accumulated in int32 from 2^21 (rounding), shifted and saturated to uint8 -- the intermediate
                   image between the passes is uint8 too. A pass whose input and output sizes agree is skipped.
  CenterCrop(c)    top = int(round((H - c) / 2.0)), left = int(round((W - c) / 2.0)) (Python round: half to even)
  ToTensor         uint8 -> float32 / 255;  ExpandChannels: the one channel three times

librdx's `rdx_transform_image` (api_transform.hip) computes the same tables in C doubles on the host and runs the two integer passes + crop + /255 on the GPU;
its output must equal this file's (and Pillow's) bit for bit."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resized_size(w: int, h: int, size: int) -> Tuple[int, int]:
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def center_crop_offsets(w: int, h: int, crop: int) -> Tuple[int, int]:
    return int(round((w - crop) / 2.0)), int(round((h - crop) / 2.0))          # (left, top)


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the whole axis (box = (0, in_size)):
    (bounds int32 [out][2] = (xmin, count), kk int32 [out][ksize])."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # (double)(in1 - in0) / outSize with float box edges
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                                                # BILINEAR.support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)                                     # C (int): truncation towards zero (the value is >= -0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            w = 1.0 - t if t < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            v = k[x] * float(1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
    return bounds, kk


def _pass_rows(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """One resampling pass along the LAST axis of a uint8 image: out[.., xx] = clip8((2^21 + sum_x img[.., xmin + x] * kk[xx][x]) >> 22)."""
    out = np.empty(img.shape[:-1] + (bounds.shape[0],), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (src[..., xmin:xmin + cnt] * kk[xx, :cnt].astype(np.int64)).sum(-1) + (1 << (PRECISION_BITS - 1))
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)            # int32 in C: no overflow (255 * 2^22 * ~1 < 2^31)
    return out


def pil_resize_bilinear(arr: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """PIL.Image.fromarray(arr, "L").resize((new_w, new_h), BILINEAR) for a uint8 [H, W] array."""
    h, w = arr.shape
    out = arr
    if new_w != w:
        out = _pass_rows(out, *precompute_coeffs(w, new_w))
    if new_h != h:
        out = np.ascontiguousarray(_pass_rows(np.ascontiguousarray(out.T), *precompute_coeffs(h, new_h)).T)
    return out


def inference_transform(arr: np.ndarray, resize: int, crop: int) -> np.ndarray:
    """uint8 [H, W] -> float32 [3, crop, crop]: Resize(resize) -> CenterCrop(crop) -> ToTensor -> ExpandChannels."""
    h, w = arr.shape
    nw, nh = resized_size(w, h, resize)
    if nw < crop or nh < crop:
        raise ValueError(f"image {nw}x{nh} after Resize({resize}) is smaller than the {crop} px crop")
    r = pil_resize_bilinear(arr, nw, nh)
    left, top = center_crop_offsets(nw, nh, crop)
    x = r[top:top + crop, left:left + crop].astype(np.float32) / np.float32(255.0)
    return np.repeat(x[None], 3, axis=0)

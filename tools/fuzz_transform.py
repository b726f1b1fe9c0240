#!/usr/bin/env python
"""One-off fuzz of rdx_transform_image against PIL.Image.resize + the crop / ToTensor rule: N random image sizes (32 .. 3200 px per side,
aspect ratios up to 4:1, both crops), every output element compared. python tools/fuzz_transform.py [N] [seed]"""
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radialog_amd import transforms as T  # noqa: E402
from radialog_amd.config import small_cfg  # noqa: E402
from radialog_amd.engine import RdxEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    eng = RdxEngine(small_cfg(), dtype="f16", device=0, max_batch=1, max_len=64, llama=False)
    bad = 0
    for i in range(n):
        short = int(rng.integers(32, 3200))
        long = min(int(short * rng.uniform(1.0, 4.0)), 4096)
        h, w = (short, long) if rng.integers(2) else (long, short)
        kind = int(rng.integers(3))
        a = (rng.integers(0, 256, (h, w), dtype=np.uint8) if kind == 0 else
             (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8) if kind == 1 else
             np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8))
        pil = Image.fromarray(a)
        for crop in (448, 488):
            host = T.create_chest_xray_transform_for_inference(512, center_crop_size=crop)(pil)
            dev = T.create_chest_xray_transform_for_inference(512, center_crop_size=crop, engine=eng)(pil)
            d = int((dev.cpu() != host).sum())
            if d:
                bad += 1
                print(f"MISMATCH {h}x{w} crop {crop}: {d} elements", flush=True)
    print(f"fuzz_transform: {n} sizes x 2 crops, {bad} mismatching outputs")
    eng.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

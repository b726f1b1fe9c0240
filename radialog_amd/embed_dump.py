"""Offline embedding dump: the batch-256 `forward_image` caller and the on-disk hand-off format between the encoder
and the decoder (pretraining/train.py:134-173 writes `{dicom: float32[32,768]}` pickles; modeling_llama_imgemb.py:454-462
reads them)."""
from __future__ import annotations

import pickle
from typing import Dict, List, Optional

import numpy as np
import torch

from .blip2_qformer import Blip2Qformer

_MODEL = {}


def dump_embeddings(images: torch.Tensor, dicoms: List[str], batch_size: int = 256, dtype: str = "bf16", device: int = 0,
                    out_path: Optional[str] = None, model: Optional[Blip2Qformer] = None, synthetic: bool = False) -> Dict[str, np.ndarray]:
    """images float32[N,3,448,448] -> {dicom: float32[32,768]}; optionally pickled to `out_path` (the file
    modeling_llama_imgemb.py:454-462 opens). `model` is the loaded Blip2Qformer (pretraining/train.py:118-131 builds it from the
    config); only `synthetic=True` builds a random-init one here."""
    if len(dicoms) != images.shape[0]:
        raise ValueError("one dicom id per image")
    if model is None:
        if not synthetic:
            raise ValueError("dump_embeddings needs a loaded Blip2Qformer (`model=`), or synthetic=True for random-init weights")
        key = (dtype, device, images.shape[-1])
        if key not in _MODEL:
            _MODEL[key] = Blip2Qformer(img_size=images.shape[-1], dtype=dtype, synthetic=True).to(torch.device("cuda", device)).eval()
        model = _MODEL[key]
    embeddings = {}
    for s in range(0, len(dicoms), batch_size):
        q = model.forward_image(images[s: s + batch_size].to(model.device))[0]
        q = q.cpu().numpy()
        for j, d in enumerate(dicoms[s: s + batch_size]):
            embeddings[d] = q[j]
    if out_path:
        with open(out_path, "wb") as f:
            pickle.dump(embeddings, f)
    return embeddings

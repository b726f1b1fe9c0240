"""Findings-classifier inference behind the reference's call surface.

Reference: `findings_classifier/chexpert_model.py:7-21` (ChexpertClassifier: BioViL-T image encoder with its stock
128-wide projector -> avg_pool2d(4) -> fc 2048 -> 512 -> 14) and its call site `demo.py:155-170,:256-261`
(`cp_model(cp_image[None].half().cuda())`, then `sigmoid(logits) > 0.5` picks the CheXpert class names that go into the
prompt). Only inference is mirrored; the Lightning training wrapper (`findings_classifier/chexpert_train.py`) is not on
the path. The forward runs in librdx (`rdx_classify_findings`): trunk and projector are the encoder's kernels, the head
is `avgpool_flatten_k` + two GEMMs. No CPU fallback.
"""
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .config import RaDialogCfg, classifier_cfg

CHEXPERT_COLS = ["No Finding", "Enlarged Cardiomediastinum", "Cardiomegaly", "Lung Opacity", "Lung Lesion", "Edema",
                 "Consolidation", "Pneumonia", "Atelectasis", "Pneumothorax", "Pleural Effusion", "Pleural Other",
                 "Fracture", "Support Devices"]                      # demo.py:157-163


class ChexpertClassifier:
    """`ChexpertClassifier(num_classes)`; `load_state_dict(sd)` takes the reference module's state_dict
    (`biovil_encoder.*`, `fc1.*`, `fc2.*`; a Lightning checkpoint's `model.` prefix is stripped); `model(x)` returns
    logits like the reference's forward. `.half()` / `.cuda()` / `.eval()` are accepted for call-site compatibility."""

    def __init__(self, num_classes: int = 14, cfg: Optional[RaDialogCfg] = None, dtype: str = "f16", device: int = 0,
                 max_batch: int = 8):
        self.cfg = cfg or classifier_cfg()
        if num_classes != self.cfg.cls.classes:
            raise ValueError(f"num_classes {num_classes} != configured {self.cfg.cls.classes}")
        self.dtype, self.device_index, self.max_batch = dtype, device, max_batch
        self.class_names = list(CHEXPERT_COLS)
        self._engine = None
        self._get: Optional[Callable[[str], torch.Tensor]] = None

    # -- weights -----------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        self.set_weight_getter(lambda name: sd[name].float())
        return self

    def set_weight_getter(self, get: Callable[[str], torch.Tensor]):
        self._get = get
        self._engine = None
        return self

    @classmethod
    def load_from_checkpoint(cls, ckpt_path: str, num_classes: int = 14, class_names=None, strict: bool = False, **kw):
        """`LitIGClassifier.load_from_checkpoint` as demo.py:165 calls it: a Lightning .ckpt with `state_dict`."""
        ck = torch.load(ckpt_path, map_location="cpu")
        m = cls(num_classes=num_classes, **kw)
        m.load_state_dict(ck.get("state_dict", ck))
        if class_names is not None:
            m.class_names = list(class_names)
        return m

    # -- call-site compatibility -------------------------------------------------------------------------------------
    def eval(self):
        return self

    def half(self):
        self.dtype = "f16"
        return self

    def cuda(self):
        return self

    def _ensure(self):
        if self._engine is None:
            from .engine import RdxEngine
            if self._get is None:
                raise RuntimeError("ChexpertClassifier: no weights loaded (load_state_dict / load_from_checkpoint)")
            eng = RdxEngine(self.cfg, dtype=self.dtype, device=self.device_index, max_batch=self.max_batch, max_len=32,
                            classifier=True)
            eng.load_weights(self._get)
            self._engine = eng
        return self._engine

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B,3,488,488] (any float dtype, any device) -> logits [B,14] on the GPU, in x's dtype."""
        eng = self._ensure()
        return eng.classify_findings(x.float()).to(x.dtype if x.is_floating_point() else torch.float32)

    forward = __call__

    def predict_findings(self, x: torch.Tensor) -> list:
        """demo.py:257-262: sigmoid > 0.5 -> ', '.join(class names).lower()."""
        probs = torch.sigmoid(self(x).float())
        names = np.asarray(self.class_names)
        return [", ".join(names[(p > 0.5).cpu().numpy()].tolist()).lower().strip() for p in probs]

"""LAVIS-style plugin surface of the image-encode half, backed by librdx.

Mirrors what demo.py:149-153 / pretraining/train.py:118-142 touch:
    cfg   = Config(args)                       # YAML with model.arch=blip2, vit_model=biovil, num_query_token=32 ...
    task  = tasks.setup_task(cfg)
    model = task.build_model(cfg)              # registry.get_model_class("blip2").from_config(cfg.model_cfg)
    model = model.to(device); model.eval()
    qformer_embs, image_embeds = model.forward_image(image)       # blip2_qformer.py:467-484
Everything numeric happens in rdx_encode_image; this file only marshals tensors and loads weights.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .config import QFormerCfg, RaDialogCfg, VisionCfg, LlamaCfg


class Registry:
    """model/lavis/common/registry.py: the two calls the hot path uses."""
    _models: Dict[str, type] = {}

    @classmethod
    def register_model(cls, name):
        def wrap(model_cls):
            cls._models[name] = model_cls
            return model_cls
        return wrap

    @classmethod
    def get_model_class(cls, name):
        if name not in cls._models:
            raise KeyError(f"model '{name}' is not registered (known: {sorted(cls._models)})")
        return cls._models[name]


registry = Registry


class _NS(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return _NS(v) if isinstance(v, dict) else v

    def get(self, k, default=None):
        v = super().get(k, default)
        return _NS(v) if isinstance(v, dict) else v


class Config:
    """model/lavis/common/config.py:16-166, reduced to what the inference path reads: a YAML file merged with
    `--options key=value` overrides (OmegaConf is not a dependency here; PyYAML is)."""

    def __init__(self, args):
        import yaml
        self.args = args
        with open(args.cfg_path) as f:
            cfg = yaml.safe_load(f) or {}
        for opt in (getattr(args, "options", None) or []):
            key, val = opt.split("=", 1)
            node = cfg
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = yaml.safe_load(val)
        self.config = _NS(cfg)

    @property
    def model_cfg(self):
        return self.config.get("model", _NS())

    @property
    def run_cfg(self):
        return self.config.get("run", _NS())


class _Task:
    def build_model(self, cfg):
        model_cfg = cfg.model_cfg
        return registry.get_model_class(model_cfg.get("arch", "blip2")).from_config(model_cfg)


class tasks:                                    # `from model.lavis import tasks; tasks.setup_task(cfg)`
    @staticmethod
    def setup_task(cfg):
        return _Task()


@registry.register_model("blip2")
class Blip2Qformer:
    """forward_image-only restatement of Blip2Qformer (blip2_qformer.py:26-89,:467-484,:630-657)."""

    def __init__(self, vit_model="biovil", img_size=448, num_query_token=32, cross_attention_freq=2, dtype="bf16",
                 cfg: Optional[RaDialogCfg] = None, max_txt_len=32, **_unused):
        if vit_model != "biovil":
            raise NotImplementedError("RaDialog only instantiates vit_model='biovil' (blip2.py:64-88)")
        base = cfg or RaDialogCfg()
        v = VisionCfg(img=img_size, stem=base.vision.stem, planes=base.vision.planes, blocks=base.vision.blocks,
                      b2v=base.vision.b2v, proj=base.vision.proj)
        q = QFormerCfg(hidden=base.qformer.hidden, layers=base.qformer.layers, heads=base.qformer.heads,
                       inter=base.qformer.inter, enc_width=v.proj, n_query=num_query_token, cross_freq=cross_attention_freq)
        self.cfg = RaDialogCfg(llama=base.llama, qformer=q, vision=v)
        self.dtype = dtype
        self.max_txt_len = max_txt_len
        self.device = torch.device("cpu")
        self._engine = None
        self._weights = None          # getter over reference-named fp32 tensors
        self.training = False

    # -- construction ----------------------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, cfg):
        model = cls(vit_model=cfg.get("vit_model", "biovil"), img_size=cfg.get("image_size", 448),
                    num_query_token=cfg.get("num_query_token", 32), cross_attention_freq=cfg.get("cross_attention_freq", 2),
                    dtype=cfg.get("dtype", "bf16"), max_txt_len=cfg.get("max_txt_len", 32))
        model.load_checkpoint_from_config(cfg)
        return model

    def load_checkpoint_from_config(self, cfg):
        """base_model.py:89-102: `load_finetuned` -> `finetuned` path, else `pretrained`. With no checkpoint reachable
        (no network here) the deterministic random-init generator stands in (`synthetic: true`)."""
        path = cfg.get("finetuned") if cfg.get("load_finetuned", False) else cfg.get("pretrained")
        if path and os.path.isfile(str(path)):
            self.load_checkpoint(path)
        elif cfg.get("synthetic", True):
            from .engine import synth_getter
            self._weights = ("synth", None)
        else:
            raise RuntimeError("checkpoint url or path is invalid")          # base_model.py:43-44

    def load_checkpoint(self, url_or_filename):
        """LAVIS checkpoint_N.pth holds {'model': state_dict} with trainable + buffer tensors only (runner_base.py:658-683);
        the frozen BioViL-T trunk comes from its own .pt (biovil_t/pretrained.py:26-32). Both may be merged by passing
        a combined state dict file."""
        if not os.path.isfile(url_or_filename):
            raise RuntimeError("checkpoint url or path is invalid")
        ck = torch.load(url_or_filename, map_location="cpu")
        sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
        self._weights = ("dict", {k: v.float() for k, v in sd.items() if torch.is_tensor(v)})
        return self

    # -- nn.Module-like surface -----------------------------------------------------------------------------------------
    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda":
            self.device = torch.device("cuda", device.index or 0)
            self._ensure_engine()
        else:
            self.device = device          # parking on "cpu" (demo.py:271) keeps the HBM-resident engine; no CPU compute path
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        self.training = False
        return self

    def _ensure_engine(self):
        if self._engine is not None:
            return
        from .engine import RdxEngine, synth_getter
        eng = RdxEngine(self.cfg, dtype=self.dtype, device=self.device.index or 0, vision=True, llama=False)
        kind, payload = self._weights or ("synth", None)
        if kind == "synth":
            get = synth_getter(self.cfg, eng.device)
        else:
            get = lambda name: payload[name].to(eng.device)          # noqa: E731
        eng.load_weights(get, vision=True, llama=False)
        self._engine = eng

    # -- the hot-path method ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_image(self, image: torch.Tensor, previous_image: Optional[torch.Tensor] = None):
        """image float32[B,3,S,S] on the model device -> (last_hidden_state f32[B,32,768], image_embeds f32[B,196,1408]).
        `previous_image` is an extension: the reference's forward_image never passes one, but its BioViL-T encoder has the
        two-image branch (MultiImageEncoder.forward, biovil_t/encoder.py:117-123) and ships the pooler weights."""
        if self._engine is None:
            raise RuntimeError("Blip2Qformer.forward_image needs the model on a GPU: call .to('cuda') first "
                               "(there is no CPU implementation of the hot path)")
        return self._engine.encode_image(image, previous_image=previous_image)

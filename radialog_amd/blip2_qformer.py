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
                 cfg: Optional[RaDialogCfg] = None, max_txt_len=32, synthetic: bool = False, **_unused):
        if vit_model != "biovil":
            raise NotImplementedError("RaDialog only instantiates vit_model='biovil' (blip2.py:64-88)")
        import dataclasses
        base = cfg or RaDialogCfg()
        v = dataclasses.replace(base.vision, img=img_size)
        q = dataclasses.replace(base.qformer, enc_width=v.proj, n_query=num_query_token, cross_freq=cross_attention_freq)
        self.cfg = RaDialogCfg(llama=base.llama, qformer=q, vision=v)
        self.dtype = dtype
        self.max_txt_len = max_txt_len
        self.device = torch.device("cpu")
        self._engine = None
        self._weights = ("synth", None) if synthetic else None     # ("dict", reference-named fp32 tensors) once loaded
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
        """base_model.py:89-102: `load_finetuned` (default True) -> `finetuned` path through load_checkpoint, else `pretrained`
        through load_from_pretrained. The frozen BioViL-T trunk is not part of those files: the reference's constructor loads it
        from `biovil_t_image_model_proj_size_128.pt` (blip2.py:80-84, pretrained.py:48-60, downloaded into tempfile.gettempdir());
        here `biovil_t_weights` in the config, $RDX_BIOVIL_T_WEIGHTS or that same temp-dir file name it. `synthetic: true` is the
        explicit opt-in to the deterministic random-init weights of benchmarks and tests; without it a missing file raises."""
        if cfg.get("synthetic", False):
            self._weights = ("synth", None)
            return
        import tempfile
        pt = cfg.get("biovil_t_weights") or os.environ.get("RDX_BIOVIL_T_WEIGHTS") or \
            os.path.join(tempfile.gettempdir(), "biovil_t_image_model_proj_size_128.pt")
        self.load_biovil_t(pt)
        if cfg.get("load_finetuned", True):
            path = cfg.get("finetuned", None)
            assert path is not None, "Found load_finetuned is True, but finetune_path is None."
            self.load_checkpoint(path)
        elif cfg.get("load_pretrained", True):
            self.load_checkpoint(cfg.get("pretrained", None))

    def _state(self) -> Dict[str, torch.Tensor]:
        if self._weights is None or self._weights[0] != "dict":
            self._weights = ("dict", {})
        return self._weights[1]

    def load_biovil_t(self, path):
        """ImageModel.__init__(pretrained_model_path=...) (biovil_t/model.py:56-65): the BioViL-T state dict (`encoder.*` trunk,
        backbone_to_vit, vit_pooler; `projector.*` of the stock 128-wide projector) with every `projector` key DROPPED -- the
        1408-wide projector of this model is not in that file -- loaded under `visual_encoder.`."""
        if not path or not os.path.isfile(str(path)):
            raise RuntimeError(f"BioViL-T image-model weights not found at {path!r} (no network access to download them)")
        sd = torch.load(str(path), map_location="cpu")
        st = self._state()
        for k, v in sd.items():
            if k.startswith("projector") or not torch.is_tensor(v):
                continue
            st["visual_encoder." + k] = v.float()
        return self

    def load_checkpoint(self, url_or_filename):
        """base_model.py:29-56 on a LAVIS `checkpoint_N.pth`: {'model': state_dict, 'optimizer': ..., 'config': ..., 'epoch': ...}
        whose state_dict holds the trainable parameters and ALL buffers only -- parameters with requires_grad=False (the frozen
        visual encoder: conv / BatchNorm affine / projector weights) were deleted before saving (runner_base.py:662-670), the
        BatchNorm running statistics were not. Loaded non-strictly over what is already there, like load_state_dict(strict=False)."""
        if not url_or_filename or not os.path.isfile(str(url_or_filename)):
            raise RuntimeError("checkpoint url or path is invalid")                   # base_model.py:43-44
        ck = torch.load(str(url_or_filename), map_location="cpu")
        sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
        st = self._state()
        for k, v in sd.items():
            if torch.is_tensor(v) and v.is_floating_point():
                st[k] = v.float()
        return self

    load_from_pretrained = load_checkpoint                                             # blip2.py:91-110: same non-strict load

    def _resolve_weights(self):
        """Getter over the assembled state dict. Every tensor of the hot path must be present; the only family no released file
        carries is the 1408-wide projector (random at construction in the reference and frozen, so never saved): those tensors
        are drawn with torch's default initialisers (modules.py:43-47: Conv2d kaiming-uniform, BatchNorm2d ones/zeros), loudly."""
        import math
        import warnings
        from . import synth
        st = dict(self._weights[1])
        v = self.cfg.vision
        J = "visual_encoder.projector.model."
        drawn = []

        def conv_default(cout, cin):
            bound = 1.0 / math.sqrt(cin)                   # kaiming_uniform_(a=sqrt(5)) on a 1x1 kernel
            return (torch.rand(cout, cin, 1, 1) * 2 - 1) * bound

        defaults = {J + "0.weight": lambda: conv_default(v.proj, 2 * v.b2v), J + "1.weight": lambda: torch.ones(v.proj),
                    J + "1.bias": lambda: torch.zeros(v.proj), J + "1.running_mean": lambda: torch.zeros(v.proj),
                    J + "1.running_var": lambda: torch.ones(v.proj), J + "3.weight": lambda: conv_default(v.proj, v.proj),
                    J + "3.bias": lambda: (torch.rand(v.proj) * 2 - 1) / math.sqrt(v.proj)}
        for k, mk in defaults.items():
            if k not in st:
                st[k] = mk()
                drawn.append(k)
        if drawn:
            warnings.warn("Blip2Qformer: no loaded file holds " + ", ".join(drawn) + " -- the reference draws the 1408-wide projector at "
                          "random when the model is constructed and never saves it (frozen); drawn here with the same initialisers "
                          "from torch's current RNG state. Pass them in the checkpoint to reproduce a specific run.")
        need = [k for k in {**synth.vision_specs(v), **synth.qformer_specs(self.cfg.qformer)} if ".vit_pooler." not in k]
        missing = [k for k in need if k not in st]
        if missing:
            raise RuntimeError(f"checkpoint lacks {len(missing)} tensors of the image-encode path, e.g. {missing[:4]}")
        has_pooler = any(".vit_pooler." in k for k in st)

        def get(name):
            if ".vit_pooler." in name and not has_pooler:
                raise KeyError(name)                       # optional two-image mode (weights.vision_items)
            return st[name]
        return get

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
        if self._weights is None:
            eng.close()
            raise RuntimeError("Blip2Qformer has no weights: from_config with a checkpoint, load_checkpoint(...), or `synthetic: true`")
        if self._weights[0] == "synth":
            get = synth_getter(self.cfg, eng.device)
        else:
            try:
                host = self._resolve_weights()
            except Exception:
                eng.close()
                raise
            get = lambda name: host(name).to(eng.device)          # noqa: E731
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

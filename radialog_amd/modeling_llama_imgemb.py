"""`LlamaForCausalLM` with the RaDialog image splice, backed by librdx (mirror of the reference's call surface).

What demo.py:221-236,:287-301 and test.py:287-365 do with the reference class keeps working:
    lang_model = LlamaForCausalLM.from_pretrained(path_or_None, torch_dtype=torch.float16, device_map='auto')
    lang_model.base_model.img_proj_layer = nn.Linear(768, hidden)        # weights arrive with the adapter file
    lang_model = PeftModelForCausalLM.from_pretrained(lang_model, lora_dir)   # here: lang_model.load_adapter(lora_dir)
    out = lang_model.generate(input_ids=ids, dicom=[...] or None, use_img=bool, return_dict_in_generate=True,
                              output_scores=True, max_new_tokens=300)
    out.sequences  int64[B, T+n]   (prompt included; rows that finished early are padded with pad id 0)
    out.scores     tuple of n tensors [B, V]
The image embeddings reach `generate` the way the reference passes them (modeling_llama_imgemb.py:454-462,:571-579):
`use_img=True` -> torch.load("current_chat_img.pt"); `dicom=[...]` -> lookup in `self.model.blip_embeddings`
(pretraining/embs/*_embeddings_{train_all,test}.pkl when present). `qformer_embs=` can also be passed directly.
"""
from __future__ import annotations

import os
import pickle
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from .config import LlamaCfg, RaDialogCfg

EMB_TEST_PKL = "pretraining/embs/stage1_pt_instruct_blip_origlr_img448_embeddings_test.pkl"
EMB_TRAIN_PKL = "pretraining/embs/stage1_pt_instruct_blip_origlr_img448_embeddings_train_all.pkl"


class GenerateOutput(SimpleNamespace):
    """`.sequences`, `.scores` like transformers' GreedySearchDecoderOnlyOutput."""


class _BaseModel:
    """`lang_model.base_model` / `lang_model.model`: holds the side-channel dict and the attribute the callers inject."""

    def __init__(self, config):
        self.config = config
        self.img_proj_layer = None
        self.blip_embeddings = {}
        try:                                                       # modeling_llama_imgemb.py:454-459
            with open(EMB_TRAIN_PKL, "rb") as f:
                self.blip_embeddings = pickle.load(f)
        except Exception:
            self.blip_embeddings = {}
        if os.path.exists(EMB_TEST_PKL):                           # (:461 opens it unguarded; absent file = demo-only use)
            with open(EMB_TEST_PKL, "rb") as f:
                self.blip_embeddings.update(pickle.load(f))

    @property
    def device(self):
        return self._owner.device


class LlamaForCausalLM:
    def __init__(self, cfg: Optional[LlamaCfg] = None, dtype: str = "bf16", max_batch: int = 12, max_len: int = 1024,
                 lora: bool = True, device: int = 0):
        self.lcfg = cfg or LlamaCfg()
        self.config = SimpleNamespace(hidden_size=self.lcfg.hidden, vocab_size=self.lcfg.vocab,
                                      num_hidden_layers=self.lcfg.layers, pad_token_id=0, eos_token_id=2)
        self.model = _BaseModel(self.config)
        self.model._owner = self
        self.base_model = self.model
        self.dtype, self.lora = dtype, lora
        self.max_batch, self.max_len = max_batch, max_len
        self.device = torch.device("cuda", device)
        self._engine = None
        self._state = None            # reference-named fp32 tensors (dict) or None -> synthetic
        self.training = False

    # -- construction -------------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, name_or_path=None, torch_dtype=torch.float16, device_map="auto", **kw):
        dt = "f16" if torch_dtype == torch.float16 else "bf16"
        self = cls(cfg=kw.pop("cfg", None), dtype=dt, **kw)
        if name_or_path and os.path.isdir(str(name_or_path)):
            self._state = _load_hf_dir(str(name_or_path))
        elif name_or_path and not kw.get("allow_synthetic", True):
            raise OSError(f"{name_or_path} is not a local directory (no network access)")
        return self

    def load_adapter(self, lora_dir: str):
        """peft adapter dir: adapter_model.bin = LoRA A/B + img_proj_layer.{weight,bias} (finetune.py:139-145)."""
        sd = torch.load(os.path.join(lora_dir, "adapter_model.bin"), map_location="cpu")
        self._state = self._state or {}
        for k, v in sd.items():
            k = k.replace("base_model.model.", "", 1).replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
            self._state[k] = v.float()
        self.lora = True
        return self

    def resize_token_embeddings(self, n):                          # test.py:297 (adds the <IMG> row)
        self.lcfg = LlamaCfg(**{**self.lcfg.__dict__, "vocab": n})
        self.config.vocab_size = n
        return self

    def half(self):
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self):
        return self

    def _ensure_engine(self):
        if self._engine is not None:
            return
        from .engine import RdxEngine, synth_getter
        cfg = RaDialogCfg(llama=self.lcfg)
        eng = RdxEngine(cfg, dtype=self.dtype, device=self.device.index or 0, max_batch=self.max_batch, max_len=self.max_len,
                        lora=self.lora, vision=False, llama=True)
        if self._state is None:
            get = synth_getter(cfg, eng.device, lora=self.lora)
        else:
            st, V = self._state, self.lcfg.vocab

            def get(name):
                t = st[name].to(eng.device)
                if name in ("model.embed_tokens.weight", "lm_head.weight") and t.shape[0] < V:   # resize_token_embeddings
                    t = torch.cat([t, t.mean(0, keepdim=True).expand(V - t.shape[0], -1)], 0)
                return t
        eng.load_weights(get, vision=False, llama=True)
        self._engine = eng

    # -- generation -------------------------------------------------------------------------------------------------------------
    def _image_embs(self, B, dicom, use_img, qformer_embs):
        if qformer_embs is not None:
            return qformer_embs
        if use_img:                                                # modeling_llama_imgemb.py:576
            e = torch.load("current_chat_img.pt")
            return e.expand(B, -1, -1) if e.shape[0] == 1 and B > 1 else e
        if dicom is not None:                                      # :579 (KeyError for an unknown id, like the reference)
            return torch.tensor(np.array([self.model.blip_embeddings[d] for d in dicom]))
        return None

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor = None, dicom: Optional[List[str]] = None, use_img: bool = False,
                 return_dict_in_generate: bool = False, output_scores: bool = False, max_new_tokens: int = 20,
                 attention_mask: Optional[torch.Tensor] = None, num_beams: int = 1, do_sample: bool = False,
                 qformer_embs: Optional[torch.Tensor] = None, eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, **_unused):
        if num_beams != 1 or do_sample:
            raise NotImplementedError("the hot path is greedy search (num_beams=1, do_sample=False), as demo.py/test.py run it")
        if input_ids.dim() != 2:
            raise ValueError("You have to specify decoder_input_ids of shape [batch, seq]")
        B, T = input_ids.shape
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        pad = self.config.pad_token_id if pad_token_id is None else pad_token_id
        embs = self._image_embs(B, dicom, use_img, qformer_embs)
        if embs is not None and tuple(embs.shape) != (B, 32, self.lcfg.qformer_dim):
            raise ValueError(f"image embeddings should be of size {(B, 32, self.lcfg.qformer_dim)}, but are {tuple(embs.shape)}")
        self._ensure_engine()
        # multi-turn chats (demo.py:277-305 re-sends the whole conversation every turn): with `reuse_prefix_kv` set on the model
        # the KV rows of the token prefix shared with the previous call are kept and only the new turn is prefilled
        toks, scores, n = self._engine.generate(input_ids, embs, max_new=max_new_tokens, eos_id=eos, pad_id=pad,
                                                mask=attention_mask, output_scores=output_scores,
                                                reuse_prefix=bool(getattr(self, "reuse_prefix_kv", False)) and attention_mask is None)
        toks = toks[:, :n].to(torch.int64)
        # HF stops as soon as every row has emitted EOS; the engine checks every 16 steps, so trim the all-pad tail
        if eos >= 0 and n > 0:
            done = (toks == eos).cumsum(1).clamp(max=1)
            if bool(done[:, -1].all()):
                n = int(done.argmax(1).max()) + 1
                toks = toks[:, :n]
        seq = torch.cat([input_ids.to(toks.device), toks], dim=1)
        if not return_dict_in_generate:
            return seq
        sc = tuple(scores[i] for i in range(n)) if output_scores else None
        return GenerateOutput(sequences=seq, scores=sc)


def _load_hf_dir(path: str):
    """Read a local HF checkpoint directory (safetensors or .bin shards) into reference-named fp32 tensors."""
    import glob
    out = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        for f in files:
            out.update({k: v.float() for k, v in load_file(f).items()})
    else:
        for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
            out.update({k: v.float() for k, v in torch.load(f, map_location="cpu").items()})
    if not out:
        raise OSError(f"no weights found under {path}")
    return out

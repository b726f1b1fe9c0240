"""`LlamaForCausalLM` with the RaDialog image splice, backed by librdx (mirror of the reference's call surface).

What demo.py:221-236,:287-301 and test.py:287-365 do with the reference class keeps working:
    lang_model = LlamaForCausalLM.from_pretrained(path_or_None, torch_dtype=torch.float16, device_map='auto')
    lang_model.base_model.img_proj_layer = nn.Linear(768, hidden)        # weights arrive with the adapter file
    lang_model = PeftModelForCausalLM.from_pretrained(lang_model, lora_dir, torch_dtype=torch.float16)    # same class name here
    out = lang_model.generate(input_ids=ids, dicom=[...] or None, use_img=bool, return_dict_in_generate=True,
                              output_scores=True, max_new_tokens=300)
    out.sequences  int64[B, T+n]   (prompt included; rows that finished early are padded with pad id 0)
    out.scores     tuple of n tensors [B, V]
The image embeddings reach `generate` the way the reference passes them (modeling_llama_imgemb.py:454-462,:571-579):
`use_img=True` -> torch.load("current_chat_img.pt"); `dicom=[...]` -> lookup in `self.model.blip_embeddings`
(pretraining/embs/*_embeddings_{train_all,test}.pkl when present). `qformer_embs=` can also be passed directly.
"""
from __future__ import annotations

import json
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
                 lora: bool = True, device: int = 0, weights_fp8: bool = False):
        self.lcfg = cfg or LlamaCfg()
        self.config = SimpleNamespace(hidden_size=self.lcfg.hidden, vocab_size=self.lcfg.vocab,
                                      num_hidden_layers=self.lcfg.layers, pad_token_id=0, eos_token_id=2)
        self.model = _BaseModel(self.config)
        self.model._owner = self
        self.base_model = self.model
        self.dtype, self.lora = dtype, lora
        self.max_batch, self.max_len = max_batch, max_len
        # BASELINE configs[4]: the decoder's GEMM weights (and, in the prefill / from batch 3, the activations) in OCP e4m3 on the fp8 MFMA. LoRA stays
        # un-merged: the LoRA-B product is a model-dtype epilogue, the 16 LoRA-A rows are part of the QKV weight and quantised with it (their own
        # row scales; weights.py -- the oracle mirrors it, so what this costs on a trained adapter, whose A rows are small against a row's absmax, is
        # NOT measured here: no adapter file is reachable offline). No reference counterpart: `from_pretrained(..., weights_fp8=True)` is this build's extra kwarg
        self.weights_fp8 = bool(weights_fp8)
        self.device = torch.device("cuda", device)
        self._engine = None
        self._state = None            # reference-named tensors (dict, stored dtype); with _synthetic they overlay the random-init generator
        self._synthetic = False       # only from_pretrained(synthetic=True) turns the deterministic random-init weights on
        self.training = False

    # -- construction -------------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, name_or_path=None, torch_dtype=torch.float16, device_map="auto", synthetic: bool = False, **kw):
        """`name_or_path` must be a local HF checkpoint directory (safetensors or pytorch_model*.bin shards). Anything else
        -- a hub id such as the reference's 'lmsys/vicuna-7b-v1.3' (there is no network here), a typo, None -- raises OSError
        like transformers does for an unreachable model, UNLESS the caller opts in to the deterministic random-init weights
        with `synthetic=True` (benchmarks and parity tests). Nothing falls back to random weights silently."""
        dt = "f16" if torch_dtype == torch.float16 else "bf16"
        kw.pop("use_ram_optimized_load", None)
        self = cls(cfg=kw.pop("cfg", None), dtype=dt, **kw)
        self._synthetic = bool(synthetic)
        if synthetic:
            self._state = None
        elif name_or_path and os.path.isdir(str(name_or_path)):
            self._state = _load_hf_dir(str(name_or_path))
            self.lora = False                      # a bare base model; PeftModelForCausalLM.from_pretrained / load_adapter adds LoRA
        else:
            raise OSError(f"{name_or_path!r} is not a local checkpoint directory (no network access; pass synthetic=True for the "
                          "deterministic random-init weights of the benchmarks)")
        return self

    def load_adapter(self, lora_dir: str):
        """peft adapter dir as finetune.py:121-150 writes it: `adapter_model.bin` = LoRA A/B of q_proj and v_proj under peft's key
        names (`base_model.model.model.layers.N.self_attn.q_proj.lora_A[.default].weight`) + `base_model.model.model.img_proj_layer.
        {weight,bias}`; `adapter_config.json` carries r, lora_alpha and target_modules (finetune.py:167-173: r=8, alpha=16,
        ["q_proj", "v_proj"]). The engine's fused LoRA epilogue supports exactly that family: anything else raises."""
        cfg_path = os.path.join(lora_dir, "adapter_config.json")
        if not os.path.isfile(cfg_path):
            raise OSError(f"{cfg_path} not found: not a peft adapter directory")
        with open(cfg_path) as f:
            acfg = json.load(f)
        r, alpha = int(acfg.get("r", 8)), float(acfg.get("lora_alpha", 16))
        targets = set(acfg.get("target_modules") or [])
        if targets != {"q_proj", "v_proj"}:
            raise ValueError(f"adapter targets {sorted(targets)}: the fused LoRA path covers q_proj and v_proj (finetune.py:171)")
        if r != 8:
            raise ValueError(f"adapter rank r={r}: the fused LoRA epilogue is built for r=8 (finetune.py:168)")
        if float(acfg.get("lora_dropout", 0.0)) and not acfg.get("inference_mode", True):
            raise ValueError("adapter_config.json is not in inference mode")
        bin_path = os.path.join(lora_dir, "adapter_model.bin")
        if not os.path.isfile(bin_path):
            raise OSError(f"{bin_path} not found")
        sd = torch.load(bin_path, map_location="cpu")
        self.lcfg = LlamaCfg(**{**self.lcfg.__dict__, "lora_r": r, "lora_alpha": alpha})
        if self._state is None and not self._synthetic:
            raise RuntimeError("load_adapter: no base weights loaded (LlamaForCausalLM.from_pretrained first)")
        self._state = {} if self._state is None else self._state
        self._adapter_keys = set()
        for k, v in sd.items():
            k = k.replace("base_model.model.", "", 1).replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
            self._state[k] = v.float()
            self._adapter_keys.add(k)
        self.lora = True
        if self._engine is not None:               # an engine built before the adapter arrived: release its weight replica and KV cache
            self._engine.close()
        self._engine = None
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

    def close(self):
        """Release the engine's weight replica and KV cache (no reference counterpart: torch frees on garbage collection; the
        entry-point tests run several models in one process)."""
        if self._engine is not None:
            self._engine.close()
        self._engine = None

    def _ensure_engine(self):
        if self._engine is not None:
            return
        from .engine import RdxEngine, synth_getter
        cfg = RaDialogCfg(llama=self.lcfg)
        eng = RdxEngine(cfg, dtype=self.dtype, device=self.device.index or 0, max_batch=self.max_batch, max_len=self.max_len,
                        lora=self.lora, vision=False, llama=True, weights_fp8=self.weights_fp8)
        if self._state is None and not self._synthetic:
            eng.close()
            raise RuntimeError("LlamaForCausalLM has no weights: build it with from_pretrained(local_dir) or from_pretrained(synthetic=True)")
        if self._synthetic:
            base, st = synth_getter(cfg, eng.device, lora=self.lora), (self._state or {})
            get = lambda name: st[name].to(eng.device) if name in st else base(name)          # noqa: E731
        else:
            st, V = self._state, self.lcfg.vocab
            need = ["model.embed_tokens.weight", "lm_head.weight", "model.norm.weight", "model.img_proj_layer.weight",
                    "model.img_proj_layer.bias"]
            if self.lora:
                need += [f"model.layers.{l}.self_attn.{m}.lora_{ab}.weight" for l in (0, self.lcfg.layers - 1)
                         for m in ("q_proj", "v_proj") for ab in "AB"]
            missing = [k for k in need if k not in st]
            if missing:
                eng.close()
                raise KeyError(f"checkpoint lacks {missing}: `img_proj_layer` and the LoRA matrices arrive with the adapter "
                               "(PeftModelForCausalLM.from_pretrained(lang_model, adapter_dir), demo.py:229-234)")

            def get(name):
                t = st[name].to(eng.device)
                if name in ("model.embed_tokens.weight", "lm_head.weight") and t.shape[0] < V:   # resize_token_embeddings
                    # transformers 4.28.1 `_get_resized_embeddings` / `_get_resized_lm_head` -> `_init_weights`: the new rows are drawn
                    # from normal(0, initializer_range = 0.02) (modeling_llama_imgemb.py:349-358). Drawn from a fixed-seed generator
                    # here so that two loads agree; the <IMG> embedding row is never read (the splice replaces it), its lm_head row
                    # gets a near-zero random logit like in the reference.
                    g = torch.Generator().manual_seed(32000 + (0 if name.startswith("model.") else 1))
                    new = torch.randn(V - t.shape[0], t.shape[1], generator=g) * 0.02
                    t = torch.cat([t, new.to(t)], 0)
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
                 pad_token_id: Optional[int] = None, length_penalty: float = 1.0, early_stopping: bool = False, **_unused):
        if do_sample:
            raise NotImplementedError("sampling is not on the path: every reference call site decodes deterministically")
        if num_beams < 1:
            raise ValueError("`num_beams` has to be an integer strictly greater than 0")
        if input_ids.dim() != 2:
            raise ValueError("You have to specify decoder_input_ids of shape [batch, seq]")
        B, T = input_ids.shape
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        pad = self.config.pad_token_id if pad_token_id is None else pad_token_id
        embs = self._image_embs(B, dicom, use_img, qformer_embs)
        if embs is not None and tuple(embs.shape) != (B, 32, self.lcfg.qformer_dim):
            raise ValueError(f"image embeddings should be of size {(B, 32, self.lcfg.qformer_dim)}, but are {tuple(embs.shape)}")
        self._ensure_engine()
        if num_beams > 1:
            return self._beam_generate(input_ids, embs, num_beams, max_new_tokens, eos, pad, attention_mask, length_penalty,
                                       early_stopping, return_dict_in_generate, output_scores)
        # multi-turn chats (demo.py:277-305 re-sends the whole conversation every turn): with `reuse_prefix_kv` set on the model
        # the KV rows of the token prefix shared with the previous call are kept and only the new turn is prefilled
        toks, scores, n = self._engine.generate(input_ids, embs, max_new=max_new_tokens, eos_id=eos, pad_id=pad,
                                                mask=attention_mask, output_scores=output_scores,
                                                reuse_prefix=bool(getattr(self, "reuse_prefix_kv", False)) and attention_mask is None)
        toks = toks[:, :n].to(torch.int64)
        # HF stops as soon as every row has emitted EOS; the engine polls every 4th step (api_llama.hip decode_loop), so trim the all-pad tail
        if eos >= 0 and n > 0:
            done = (toks == eos).cumsum(1).clamp(max=1)
            if bool(done[:, -1].all()):
                n = int(done.argmax(1).max()) + 1
                toks = toks[:, :n]
        seq = torch.cat([input_ids.to(toks.device), toks], dim=1)
        if not return_dict_in_generate:
            return seq
        # fresh tensors like HF's: the engine reuses its score buffer on the next call
        sc = tuple(scores[i].clone() for i in range(n)) if output_scores else None
        return GenerateOutput(sequences=seq, scores=sc)


def _beam_generate(self, input_ids, embs, num_beams, max_new_tokens, eos, pad, attention_mask, length_penalty, early_stopping,
                   return_dict_in_generate, output_scores):
    """Tail of BeamSearchScorer.finalize (transformers 4.28.1): one hypothesis per prompt, EOS appended where a hypothesis ended
    before the longest one, rows padded to a common length min(longest + 1, T + max_new_tokens)."""
    if input_ids.shape[0] * num_beams > self.max_batch:
        raise ValueError(f"batch {input_ids.shape[0]} x num_beams {num_beams} exceeds max_batch {self.max_batch} of this model")
    toks, lens, seq_scores, step_scores, n = self._engine.beam_search(input_ids, embs, num_beams, max_new_tokens, eos_id=eos, pad_id=pad,
                                                                      mask=attention_mask, length_penalty=length_penalty,
                                                                      early_stopping=early_stopping, output_scores=output_scores)
    B, T = input_ids.shape
    sent_max = min(int(lens.max()) + 1, max_new_tokens)
    gen = torch.full((B, sent_max), pad, dtype=torch.int64)
    for b in range(B):
        L = int(lens[b])
        gen[b, :L] = toks[b, :L].to(torch.int64)
        if L < sent_max and eos >= 0:
            gen[b, L] = eos
    seq = torch.cat([input_ids.to(torch.int64).cpu(), gen], dim=1).to(self.device)
    if not return_dict_in_generate:
        return seq
    sc = tuple(step_scores[i].clone() for i in range(step_scores.shape[0])) if output_scores else None
    return GenerateOutput(sequences=seq, scores=sc, sequences_scores=seq_scores.to(self.device))


LlamaForCausalLM._beam_generate = _beam_generate


class PeftModelForCausalLM:
    """`PeftModelForCausalLM.from_pretrained(lang_model, adapter_dir, torch_dtype=torch.float16[, use_ram_optimized_load=False])`
    exactly as demo.py:232-234 / test.py:301 call it (peft@e536616): loads the adapter into `lang_model` and returns an object
    with the wrapped model's surface (`generate`, `eval`, `half`, `base_model.model` = the LlamaForCausalLM)."""

    def __init__(self, model: "LlamaForCausalLM"):
        self.base_model = SimpleNamespace(model=model)

    @classmethod
    def from_pretrained(cls, model: "LlamaForCausalLM", model_id: str, torch_dtype=None, **_kw):
        model.load_adapter(str(model_id))
        return cls(model)

    def __getattr__(self, name):
        return getattr(self.base_model.model, name)

    def __setattr__(self, name, value):
        # peft's wrapper is an nn.Module whose attributes callers set on the wrapper (demo.py sets `reuse_prefix_kv`); everything but the
        # wrapper's own `base_model` belongs to the wrapped LlamaForCausalLM, where generate() reads it
        if name == "base_model":
            object.__setattr__(self, name, value)
        else:
            setattr(self.base_model.model, name, value)

    def half(self):
        return self

    def eval(self):
        self.base_model.model.eval()
        return self

    def generate(self, **kw):
        return self.base_model.model.generate(**kw)


def _load_hf_dir(path: str):
    """Read a local HF checkpoint directory (safetensors or .bin shards) into reference-named tensors IN THE STORED DTYPE (Vicuna-7B:
    fp16, 13.5 GB): nothing is inflated to fp32 on the host -- the engine widens tensor by tensor on the device (rdx_set_weight_typed)."""
    import glob
    out = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        for f in files:
            out.update(load_file(f))
    else:
        for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
            out.update(torch.load(f, map_location="cpu"))
    if not out:
        raise OSError(f"no weights found under {path}")
    return out

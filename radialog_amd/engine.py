"""RdxEngine: the thin Python owner of one librdx context (one per process and GPU).

PyTorch-ROCm is used only as the tensor container (device memory, dtype views) and for `torch.distributed`;
all arithmetic of the hot path runs in librdx's HIP kernels. There is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import torch

from . import _lib
from ._lib import RdxConfig, check
from .config import RaDialogCfg
from . import weights as W

_TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}
_RDX_DT = {"f16": _lib.RDX_DTYPE_F16, "bf16": _lib.RDX_DTYPE_BF16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class RdxEngine:
    def __init__(self, cfg: RaDialogCfg, dtype: str = "bf16", device: int = 0, max_batch: int = 1, max_len: int = 512,
                 lora: bool = True, vision: bool = True, llama: bool = True, classifier: bool = False,
                 weights_fp8: bool = False):
        if dtype not in _RDX_DT:
            raise ValueError(f"dtype must be 'f16' or 'bf16', got {dtype!r}")
        self.lib = _lib.load()                       # raises RdxLibraryError when the HIP library is absent
        if not torch.cuda.is_available():
            raise _lib.RdxError("RdxEngine needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.cfg, self.dtype, self.tdtype = cfg, dtype, _TORCH_DT[dtype]
        self.device = torch.device("cuda", device)
        self.lora = lora
        self.weights_fp8 = weights_fp8               # decoder GEMM weights also quantised to e4m3 + per-row scale (configs[4])
        l, q, v = cfg.llama, cfg.qformer, cfg.vision
        rc = RdxConfig()
        rc.dtype = _RDX_DT[dtype]
        rc.vocab, rc.hidden, rc.inter, rc.layers, rc.heads, rc.max_pos = l.vocab, l.hidden, l.inter, l.layers, l.heads, l.max_pos
        rc.rms_eps = l.rms_eps
        rc.lora_r, rc.lora_scale = (l.lora_r if lora else 0), l.lora_scale
        rc.qformer_dim = l.qformer_dim
        rc.q_hidden, rc.q_layers, rc.q_heads, rc.q_inter = q.hidden, q.layers, q.heads, q.inter
        rc.q_enc_width, rc.q_nquery, rc.q_cross_freq, rc.q_ln_eps = q.enc_width, q.n_query, q.cross_freq, q.ln_eps
        rc.v_img, rc.v_stem, rc.v_b2v, rc.v_proj, rc.v_ln_eps = v.img, v.stem, v.b2v, v.proj, v.ln_eps
        for i in range(4):
            rc.v_planes[i] = v.planes[i]
            rc.v_blocks[i] = v.blocks[i]
        rc.max_batch, rc.max_len = max_batch, max_len
        if classifier:                                   # a findings-classifier context: trunk + its own projector + head
            vision = llama = False
        rc.enable_vision, rc.enable_llama = int(vision), int(llama)
        rc.enable_cls, rc.cls_hidden, rc.cls_classes, rc.cls_pool = int(classifier), cfg.cls.hidden, cfg.cls.classes, cfg.cls.pool
        self.classifier = classifier
        self.max_batch, self.max_len = max_batch, max_len
        self.ctx = C.c_void_p()
        rcode = self.lib.rdx_create(C.byref(self.ctx), device, C.byref(rc))
        if rcode != 0:
            msg = self.lib.rdx_last_error(None)
            raise _lib.RdxError(f"rdx_create failed ({rcode}): {msg.decode() if msg else '?'}")
        self._finalized = False
        self._keep = {}

    # ------------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.rdx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.ctx, self.lib.rdx_sync(self.ctx), "rdx_sync")

    # ------------------------------------------------------------------------------------------------------------
    def _upload(self, items):
        src = {torch.float32: _lib.RDX_SRC_F32, torch.float16: _lib.RDX_SRC_F16, torch.bfloat16: _lib.RDX_SRC_BF16}
        for name, t, kind in items:
            # fp16 / bf16 tensors (a released checkpoint is stored that way) go to the library as they are: widened on the device,
            # tensor by tensor (rdx_set_weight_typed), instead of through an fp32 copy made here
            t = t.to(device=self.device).contiguous()
            if t.dtype not in src:
                t = t.to(torch.float32)
            if t.dim() == 1:
                t = t.view(1, -1)
            rows, cols = t.shape[0], t.numel() // t.shape[0]
            torch.cuda.synchronize(self.device)
            check(self.ctx, self.lib.rdx_set_weight_typed(self.ctx, name.encode(), _ptr(t), src[t.dtype], rows, cols, kind),
                  f"rdx_set_weight({name})")
            del t

    def load_weights(self, get: Callable[[str], torch.Tensor], vision: bool = True, llama: bool = True):
        """`get(name)` returns the fp32 reference-named tensor (any device). Uploads and finalizes."""
        if self.classifier:
            with torch.no_grad():
                self._upload(W.classifier_items(get, self.cfg.vision, self.cfg.cls))
            check(self.ctx, self.lib.rdx_finalize_weights(self.ctx), "rdx_finalize_weights")
            self._finalized = True
            return
        with torch.no_grad():
            if vision:
                self._upload(W.vision_items(get, self.cfg.vision))
                self._upload(W.qformer_items(get, self.cfg.qformer))
            if llama:
                self._upload(W.llama_items(get, self.cfg.llama, self.lora, fp8=self.weights_fp8, dtype=self.tdtype))
        check(self.ctx, self.lib.rdx_finalize_weights(self.ctx), "rdx_finalize_weights")
        self._finalized = True

    # ------------------------------------------------------------------------------------------------------------
    def transform_image(self, image_u8, resize: int = 512, crop: int = 448) -> torch.Tensor:
        """rdx_transform_image: the reference's inference transform on the GPU. image_u8: uint8 [H, W] (torch tensor or numpy array: the "L" image of
        demo.py:205-218) -> float32 [3, crop, crop] on this device, bit for bit what Resize(resize) -> CenterCrop(crop) -> ToTensor -> ExpandChannels give
        on the PIL image (ReportDataset.py:96-106)."""
        t = torch.as_tensor(image_u8)
        if t.dim() != 2 or t.dtype != torch.uint8:
            raise ValueError(f"Expected input of shape [1, H, W], found {tuple(t.shape)} ({t.dtype}): one 8-bit channel")
        t = t.to(self.device).contiguous()
        out = torch.empty(3, crop, crop, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_transform_image(self.ctx, _ptr(t), t.shape[0], t.shape[1], int(resize), int(crop), _ptr(out)), "rdx_transform_image")
        self.sync()
        return out

    def encode_image(self, image: torch.Tensor, want_image_embeds: bool = True, previous_image: Optional[torch.Tensor] = None):
        """image float32[B,3,S,S] on this device -> (qformer_out f32[B,nq,Hq], image_embeds f32[B,P,C] or None).
        `previous_image` (same shape) selects the BioViL-T two-image branch (ViT pooler difference features)."""
        v, q = self.cfg.vision, self.cfg.qformer
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != v.img or image.shape[3] != v.img:
            raise ValueError(f"expected image [B,3,{v.img},{v.img}], got {tuple(image.shape)}")
        image = image.to(device=self.device, dtype=torch.float32).contiguous()
        B = image.shape[0]
        out = torch.empty(B, q.n_query, q.hidden, dtype=torch.float32, device=self.device)
        emb = torch.empty(B, v.n_patches, v.proj, dtype=torch.float32, device=self.device) if want_image_embeds else None
        prev = None
        if previous_image is not None:
            if previous_image.shape != image.shape:
                raise AssertionError("current_image and previous_image shapes do not match")     # biovil_t/encoder.py:118
            prev = previous_image.to(device=self.device, dtype=torch.float32).contiguous()
        torch.cuda.synchronize(self.device)
        if prev is None:
            check(self.ctx, self.lib.rdx_encode_image(self.ctx, _ptr(image), B, _ptr(out), _ptr(emb)), "rdx_encode_image")
        else:
            check(self.ctx, self.lib.rdx_encode_image2(self.ctx, _ptr(image), _ptr(prev), B, _ptr(out), _ptr(emb)), "rdx_encode_image2")
        self.sync()
        return out, emb

    def _reusable_prefix(self, ids: torch.Tensor, qf, pad_id: int) -> int:
        """Number of leading cache slots of the previous generate(reuse_prefix=True) call that this prompt can keep: the
        longest token prefix all rows share with what was fed then (prompt + consumed answer tokens), provided the image
        embeddings are the same, the whole <IMG> block lies inside it and the rest holds neither padding nor <IMG>."""
        conv = getattr(self, "_conv", None)
        if conv is None or conv["seq"].shape[0] != ids.shape[0]:
            return 0
        if (qf is None) != (conv["qf"] is None) or (qf is not None and not torch.equal(qf, conv["qf"])):
            return 0
        seq, new = conv["seq"], ids.cpu().to(torch.int64)
        m = min(seq.shape[1], new.shape[1] - 1)                  # at least one token must be run to get logits
        if m <= 0:
            return 0
        same = (seq[:, :m] == new[:, :m]).all(dim=0).to(torch.int64)
        p = int(same.cumprod(0).sum())
        tail = new[:, p:]
        if p <= 0 or bool((tail == pad_id).any()) or bool((tail == 32000).any()):
            return 0
        return p

    def generate(self, ids: torch.Tensor, qformer_embs: Optional[torch.Tensor], max_new: int, eos_id: int = 2,
                 pad_id: int = 0, mask: Optional[torch.Tensor] = None, output_scores: bool = False, use_graph: bool = True,
                 reuse_prefix: bool = False):
        """Greedy generation. Returns (tokens int32[B,n_steps], scores [n_steps,B,V] model dtype or None, n_steps).
        reuse_prefix: multi-turn conversations -- keep the KV rows of the token prefix this prompt shares with the previous
        reuse_prefix call and prefill only the rest (rdx_generate_append); outputs are those of the full prompt."""
        B, T = ids.shape
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        m32 = None if mask is None else mask.to(device=self.device, dtype=torch.int32).contiguous()
        qf = None if qformer_embs is None else qformer_embs.to(device=self.device, dtype=torch.float32).contiguous()
        key = ("tok", B, max_new)
        toks = self._keep.get(key)
        if toks is None:
            toks = torch.zeros(B, max_new, dtype=torch.int32, device=self.device)
            self._keep[key] = toks
        toks.fill_(pad_id)
        scores = None
        if output_scores:
            skey = ("scores", B, max_new)
            scores = self._keep.get(skey)
            if scores is None:
                scores = torch.zeros(max_new, B, self.cfg.llama.vocab, dtype=self.tdtype, device=self.device)
                self._keep[skey] = scores
        n = C.c_int(0)
        torch.cuda.synchronize(self.device)
        keep = self._reusable_prefix(ids, qf, pad_id) if (reuse_prefix and mask is None) else 0
        self.last_kept_prefix = keep
        if keep:
            tail = ids32[:, keep:].contiguous()
            check(self.ctx, self.lib.rdx_generate_append(self.ctx, _ptr(tail), B, T - keep, keep, max_new, eos_id, pad_id, _ptr(toks),
                                                         _ptr(scores), C.byref(n), int(use_graph)), "rdx_generate_append")
        else:
            check(self.ctx, self.lib.rdx_generate(self.ctx, _ptr(ids32), _ptr(m32), B, T, _ptr(qf), max_new, eos_id, pad_id,
                                                  _ptr(toks), _ptr(scores), C.byref(n), int(use_graph)), "rdx_generate")
        if reuse_prefix and mask is None:
            # what the cache now holds: the prompt and every answer token that was fed back (all but the last one selected)
            self._conv = {"seq": torch.cat([ids.cpu().to(torch.int64), toks[:, :max(n.value - 1, 0)].cpu().to(torch.int64)], dim=1),
                          "qf": None if qf is None else qf.clone()}
        else:
            self._conv = None
        return toks, scores, n.value

    def beam_search(self, ids: torch.Tensor, qformer_embs: Optional[torch.Tensor], num_beams: int, max_new: int, eos_id: int = 2,
                    pad_id: int = 0, mask: Optional[torch.Tensor] = None, length_penalty: float = 1.0, early_stopping: bool = False,
                    output_scores: bool = False):
        """transformers 4.28.1 beam search over rdx_beam_search. ids int[B, T] (NOT expanded; expanded here with repeat_interleave
        like _expand_inputs_for_generation). Returns (tokens int32[B, max_new] on the host, lengths int32[B], sequence scores f32[B],
        step scores [n, B * num_beams, V] model dtype or None, n forwards)."""
        B, T = ids.shape
        k = int(num_beams)
        rep = lambda t, dt: None if t is None else t.to(device=self.device, dtype=dt).repeat_interleave(k, dim=0).contiguous()   # noqa: E731
        ids32, m32, qf = rep(ids, torch.int32), rep(mask, torch.int32), rep(qformer_embs, torch.float32)
        toks = torch.zeros(B, max_new, dtype=torch.int32)
        lens = torch.zeros(B, dtype=torch.int32)
        seq_scores = torch.zeros(B, dtype=torch.float32)
        sc = torch.zeros(max_new, B * k, self.cfg.llama.vocab, dtype=self.tdtype, device=self.device) if output_scores else None
        n = C.c_int(0)
        self._conv = None
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_beam_search(self.ctx, _ptr(ids32), _ptr(m32), B, k, T, _ptr(qf), max_new, eos_id, pad_id,
                                                 float(length_penalty), int(bool(early_stopping)), C.c_void_p(toks.data_ptr()),
                                                 C.c_void_p(lens.data_ptr()), C.c_void_p(seq_scores.data_ptr()), _ptr(sc), C.byref(n)),
              "rdx_beam_search")
        return toks, lens, seq_scores, (None if sc is None else sc[: n.value]), n.value

    def prefill(self, ids, qformer_embs, max_new, eos_id=2, pad_id=0, mask=None, want_logits=True):
        B, T = ids.shape
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        m32 = None if mask is None else mask.to(device=self.device, dtype=torch.int32).contiguous()
        qf = None if qformer_embs is None else qformer_embs.to(device=self.device, dtype=torch.float32).contiguous()
        toks = torch.full((B, max_new), pad_id, dtype=torch.int32, device=self.device)
        logits = torch.empty(B, self.cfg.llama.vocab, dtype=self.tdtype, device=self.device) if want_logits else None
        self._keep["prefill"] = (ids32, m32, qf, toks)
        self._conv = None                            # the KV cache is overwritten: nothing of an earlier conversation can be reused
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_prefill(self.ctx, _ptr(ids32), _ptr(m32), B, T, _ptr(qf), max_new, eos_id, pad_id,
                                             _ptr(toks), _ptr(logits)), "rdx_prefill")
        self.sync()
        return toks, logits

    def decode_step(self, want_logits=True, input_ids: Optional[torch.Tensor] = None):
        """One decode step on the token the previous step selected, or on caller-supplied `input_ids` int[B] (rdx_decode_step_ids)."""
        ids32, m32, qf, toks = self._keep["prefill"]
        B = ids32.shape[0]
        logits = torch.empty(B, self.cfg.llama.vocab, dtype=self.tdtype, device=self.device) if want_logits else None
        if input_ids is None:
            check(self.ctx, self.lib.rdx_decode_step(self.ctx, _ptr(logits)), "rdx_decode_step")
        else:
            forced = input_ids.to(device=self.device, dtype=torch.int32).contiguous().view(-1)
            if forced.numel() != B:
                raise ValueError(f"input_ids must hold one id per row ({B}), got {forced.numel()}")
            torch.cuda.synchronize(self.device)
            check(self.ctx, self.lib.rdx_decode_step_ids(self.ctx, _ptr(forced), _ptr(logits)), "rdx_decode_step_ids")
        self.sync()
        return toks, logits

    def kv_read(self, layer: int, which: int, batch: int) -> torch.Tensor:
        l = self.cfg.llama
        out = torch.empty(self.max_batch, l.heads, self.max_len, l.head_dim, dtype=self.tdtype, device=self.device)
        check(self.ctx, self.lib.rdx_kv_read(self.ctx, layer, which, _ptr(out)), "rdx_kv_read")
        self.sync()
        return out[:batch]

    def gemm_test(self, x, w, bias=None, resid=None, epi=0, norm_w=None, eps=1e-6, force=0):
        """out = epilogue(x @ w.T) through the production GEMM kernels. x [M,K] model dtype, w [N,K] fp32."""
        M, K = x.shape
        N = w.shape[0]
        x = x.to(self.device, self.tdtype).contiguous()
        w = w.to(self.device, torch.float32).contiguous()
        b = None if bias is None else bias.to(self.device, torch.float32).contiguous()
        r = None if resid is None else resid.to(self.device, self.tdtype).contiguous()
        nw = None if norm_w is None else norm_w.to(self.device, self.tdtype).contiguous()
        out = torch.empty(M, N // 2 if epi == 4 else N, dtype=self.tdtype, device=self.device)
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_gemm_test(self.ctx, _ptr(x), _ptr(w), _ptr(b), _ptr(r), _ptr(out), M, N, K, epi, _ptr(nw),
                                               eps, force), "rdx_gemm_test")
        return out

    def quant_test(self, x, groups=1, norm_w=None, eps=1e-6):
        """The fp8 path's activation quantisers on caller data (rdx_quant_test): x [M, K] model dtype -> (e4m3 codes uint8 [M, K], fp32 scales
        [M, groups]); norm_w given: RMSNorm -> e4m3 with one scale per row (the prefill's rmsnorm -> fp8)."""
        M, K = x.shape
        x = x.to(self.device, self.tdtype).contiguous()
        nw = None if norm_w is None else norm_w.to(self.device, self.tdtype).contiguous()
        g = 1 if norm_w is not None else groups
        out8 = torch.empty(M, K, dtype=torch.uint8, device=self.device)
        sc = torch.empty(M, g, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_quant_test(self.ctx, _ptr(x), M, K, g, 0 if norm_w is None else 1, _ptr(nw), eps, _ptr(out8), _ptr(sc)), "rdx_quant_test")
        return out8, sc

    def set_option(self, name: str, value: int):
        """rdx_set_option: "flash_min" (batched prefill attention kernel choice), "pconv" (packed / row-major encoder kernels)."""
        check(self.ctx, self.lib.rdx_set_option(self.ctx, name.encode(), int(value)), "rdx_set_option")

    def conv_test(self, x, w, bias=None, resid=None, ksize=1, stride=1, epi=0, path=0, iters=0):
        """One NHWC convolution (rdx_conv_test): x [B,H,H,Cin] model dtype, w [Cout, ksize*ksize*Cin] fp32 in the (kh, kw, c) K order;
        path 0 = row-major production dispatch, 1 = fragment-packed pconv_k, 2 = pconv_k with row-major output. Returns out
        [B,Ho,Ho,Cout] (and ms per launch when iters > 0)."""
        B, H, _, Cin = x.shape
        Cout = w.shape[0]
        Ho = (H + 2 * (ksize // 2) - ksize) // stride + 1
        x = x.to(self.device, self.tdtype).contiguous()
        w = w.to(self.device, torch.float32).contiguous()
        b = None if bias is None else bias.to(self.device, torch.float32).contiguous()
        r = None if resid is None else resid.to(self.device, self.tdtype).contiguous()
        out = torch.empty(B, Ho, Ho, Cout, dtype=self.tdtype, device=self.device)
        ms = C.c_float(0)
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_conv_test(self.ctx, _ptr(x), _ptr(w), _ptr(b), _ptr(r), _ptr(out), B, H, Cin, Cout, ksize, stride, epi,
                                               path, iters, C.byref(ms) if iters else None), "rdx_conv_test")
        return (out, ms.value) if iters else out

    def logits_test(self, x, w, n_valid=None, fp8=False):
        """(logits [M,N] model dtype, argmax int32[M]) of x @ w.T through the lm_head epilogue of the weight-streaming kernels."""
        import ctypes
        M, K = x.shape
        N = w.shape[0]
        x = x.to(self.device, self.tdtype).contiguous()
        w = w.to(self.device, torch.float32).contiguous()
        out = torch.zeros(M, N, dtype=self.tdtype, device=self.device)
        am = (ctypes.c_int32 * M)()
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_logits_test(self.ctx, _ptr(x), _ptr(w), M, N, N if n_valid is None else n_valid, K, _ptr(out),
                                                 ctypes.cast(am, ctypes.c_void_p), int(fp8)), "rdx_logits_test")
        return out, torch.tensor(list(am), dtype=torch.int32)

    def classify_findings(self, image: torch.Tensor) -> torch.Tensor:
        """ChexpertClassifier.forward: float32[B,3,S,S] on the device -> float32[B,classes] logits."""
        image = image.to(self.device, torch.float32).contiguous()
        B = image.shape[0]
        out = torch.empty(B, self.cfg.cls.classes, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_classify_findings(self.ctx, _ptr(image), B, _ptr(out)), "rdx_classify_findings")
        return out

    # -- data-parallel collective (RCCL inside the C ABI) ---------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        """128 opaque bytes from ncclGetUniqueId (rank 0); hand them to every rank, then comm_init everywhere."""
        buf = C.create_string_buffer(128)
        rc = self.lib.rdx_comm_unique_id(buf)
        if rc != 0:
            msg = self.lib.rdx_last_error(None)
            raise _lib.RdxError(f"rdx_comm_unique_id failed ({rc}): {msg.decode() if msg else '?'}")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        buf = C.create_string_buffer(unique_id, 128)
        check(self.ctx, self.lib.rdx_comm_init(self.ctx, buf, rank, world), "rdx_comm_init")

    @property
    def comm_world(self) -> int:
        """Ranks of the RCCL communicator inside librdx (0: none, or switched off by the launcher with `comm_off = True` after a
        bring-up that did not succeed on every rank)."""
        return 0 if getattr(self, "comm_off", False) else int(self.lib.rdx_comm_world(self.ctx))

    def allgather_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """int32[B_local, N] on this device -> int32[world * B_local, N], rank-major: one ncclAllGather on the engine's stream."""
        if tokens.dtype != torch.int32 or not tokens.is_cuda or tokens.dim() != 2:
            raise ValueError("allgather_tokens takes an int32 [rows, n] tensor on the engine's device")
        world = self.comm_world
        if world <= 0:
            raise _lib.RdxError("allgather_tokens: no communicator (comm_init / shard.init_comm first)")
        tokens = tokens.contiguous()
        out = torch.empty(world * tokens.shape[0], tokens.shape[1], dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)
        check(self.ctx, self.lib.rdx_allgather_tokens(self.ctx, _ptr(tokens), _ptr(out), tokens.shape[0], tokens.shape[1]), "rdx_allgather_tokens")
        self.sync()
        return out

    def kernel_bench(self, rows, N, K, H=0, ksize=0, stride=1, epi=0, iters=20, trace_wgs=0):
        """ms per launch of one GEMM / NHWC convolution through the encoder's dispatch (rdx_kernel_bench); with trace_wgs also the
        int64 [trace_wgs, 8] per-workgroup timeline of gemm_dma_k (100 MHz ticks)."""
        ms = C.c_float(0)
        tr = torch.zeros(max(trace_wgs, 1), 8, dtype=torch.int64)
        check(self.ctx, self.lib.rdx_kernel_bench(self.ctx, rows, N, K, H, ksize, stride, epi, iters, C.byref(ms),
                                                  C.c_void_p(tr.data_ptr()) if trace_wgs else None, trace_wgs), "rdx_kernel_bench")
        return (ms.value, tr) if trace_wgs else ms.value

    def time_unit(self, what: int, iters: int) -> float:
        ms = C.c_float(0)
        check(self.ctx, self.lib.rdx_time(self.ctx, what, iters, C.byref(ms)), "rdx_time")
        return ms.value


    def attn_trace(self, layer: int):
        """Debug: 8 timestamps (100 MHz ticks) inside the stand-alone decode-attention kernel (workgroup 0,0)."""
        buf = torch.zeros(8, dtype=torch.int64)
        check(self.ctx, self.lib.rdx_attn_trace(self.ctx, layer, buf.data_ptr()), "rdx_attn_trace")
        return buf

    def gemv_trace(self, what: int, layer: int, max_tiles: int = 2048):
        """Debug: per-workgroup timestamps [tiles, 8] of one stand-alone decode GEMV (1 gate/up, 2 qkv, 4 down)."""
        buf = torch.zeros(max_tiles, 8, dtype=torch.int64)
        check(self.ctx, self.lib.rdx_gemv_trace(self.ctx, what, layer, buf.data_ptr(), max_tiles), "rdx_gemv_trace")
        return buf


def synth_getter(cfg: RaDialogCfg, device, lora: bool = True) -> Callable[[str], torch.Tensor]:
    """Lazy deterministic random-init weights (radialog_amd.synth), generated on `device` one tensor at a time."""
    from . import synth
    specs: Dict[str, tuple] = {}
    specs.update(synth.vision_specs(cfg.vision))
    specs.update(synth.qformer_specs(cfg.qformer))
    specs.update(synth.llama_specs(cfg.llama, lora=lora))
    specs.update(synth.classifier_specs(cfg.vision, cfg.cls))       # `biovil_encoder.*`, fc1, fc2 (classifier contexts)

    def get(name: str) -> torch.Tensor:
        shape, gen = specs[name]
        return gen(name, shape, device)

    return get

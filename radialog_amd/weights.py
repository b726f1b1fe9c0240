"""Reference state_dict -> librdx engine tensors (load-time weight preparation).

Input is a getter `get(name) -> float32 tensor` over the REFERENCE's parameter names: a dict of
released weights, or the lazy synthetic generator of `radialog_amd.synth`. Output is a stream of
`(engine_name, float32 2-D tensor, kind)` that `RdxEngine` uploads with `rdx_set_weight`, one tensor at a time (the
fp32 Vicuna-7B set would be 26 GB at once). Everything here is exact algebra on weights, done once at load:

  * eval-mode BatchNorm folded into the preceding conv:  w' = w * g/sqrt(var+eps),  b' = beta - mean * g/sqrt(var+eps)
    (torchvision Bottleneck conv->bn pairs behind biovil_t/resnet.py:34-42; projector BN, biovil_t/modules.py:43-47)
  * conv weights re-ordered to the implicit-GEMM K order (kh, kw, c) of an NHWC activation; the 7x7 stem padded to
    7 x 8 x 4 so one 16-byte load covers two pixels
  * `missing_previous_emb` (the constant second half of the projector input, biovil_t/encoder.py:128-130) folded into
    the bias of the projector's first conv
  * q/k/v (+ the LoRA A matrices as 16 extra rows) fused into one QKV weight; gate/up interleaved 8+8 rows per
    16-row MFMA tile so SwiGLU is a tile-local epilogue; all six cross-attention K/V projections of the Q-Former
    fused into one GEMM over the image tokens
  * LayerNorm(query_tokens) (Qformer.py:103-108 with input_ids=None) is input independent -> precomputed
  * RoPE cos/sin tables built with torch exactly as LlamaRotaryEmbedding does (modeling_llama_imgemb.py:99-109)
"""
from __future__ import annotations

from typing import Callable, Iterator, Tuple

import torch
import torch.nn.functional as F

from ._lib import RDX_W_F32, RDX_W_GEMM, RDX_W_GEMM_FP8, RDX_W_TENSOR
from .config import LlamaCfg, QFormerCfg, VisionCfg

Getter = Callable[[str], torch.Tensor]
Item = Tuple[str, torch.Tensor, int]


def _bn_fold(get: Getter, conv_w: torch.Tensor, bn: str, eps: float):
    scale = get(bn + ".weight") / torch.sqrt(get(bn + ".running_var") + eps)
    bias = get(bn + ".bias") - get(bn + ".running_mean") * scale
    return conv_w * scale.view(-1, *([1] * (conv_w.dim() - 1))), bias


def _khwc(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin] with K ordered (kh, kw, c)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def vision_items(get: Getter, v: VisionCfg, prefix: str = "visual_encoder.", with_ln: bool = True) -> Iterator[Item]:
    """`prefix` = attribute name of the ImageModel in the owning module: `visual_encoder.` in Blip2Qformer
    (blip2_qformer.py), `biovil_encoder.` in ChexpertClassifier (chexpert_model.py:10), which has no ln_vision."""
    P = prefix + "encoder.encoder."
    w, b = _bn_fold(get, get(P + "conv1.weight"), P + "bn1", v.bn_eps)           # [stem,3,7,7]
    w = F.pad(w.permute(0, 2, 3, 1), (0, 1, 0, 1))                                # [stem,7,8,4], zero kw=7 / c=3
    yield "v.conv1.w", w.reshape(v.stem, 7 * 8 * 4).contiguous(), RDX_W_GEMM
    yield "v.conv1.b", b.view(1, -1), RDX_W_F32
    for li, nblk in enumerate(v.blocks, start=1):
        for blk in range(nblk):
            pre, out = f"{P}layer{li}.{blk}.", f"v.l{li}.{blk}."
            for i in (1, 2, 3):
                w, b = _bn_fold(get, get(pre + f"conv{i}.weight"), pre + f"bn{i}", v.bn_eps)
                yield out + f"c{i}.w", _khwc(w), RDX_W_GEMM
                yield out + f"c{i}.b", b.view(1, -1), RDX_W_F32
            if blk == 0:
                w, b = _bn_fold(get, get(pre + "downsample.0.weight"), pre + "downsample.1", v.bn_eps)
                yield out + "ds.w", _khwc(w), RDX_W_GEMM
                yield out + "ds.b", b.view(1, -1), RDX_W_F32
    E = prefix + "encoder."
    yield "v.b2v.w", get(E + "backbone_to_vit.weight").reshape(v.b2v, v.trunk_out).contiguous(), RDX_W_GEMM
    J = prefix + "projector.model."
    w0 = get(J + "0.weight").reshape(v.proj, 2 * v.b2v)
    miss = get(E + "missing_previous_emb").reshape(v.b2v)
    const = w0[:, v.b2v:] @ miss                                                   # contribution of the constant half
    scale = get(J + "1.weight") / torch.sqrt(get(J + "1.running_var") + v.bn_eps)
    bias = (const - get(J + "1.running_mean")) * scale + get(J + "1.bias")
    yield "v.proj1.w", (w0[:, : v.b2v] * scale[:, None]).contiguous(), RDX_W_GEMM
    yield "v.proj1.b", bias.view(1, -1), RDX_W_F32
    # optional two-image mode (VisionTransformerPooler, biovil_t/transformer.py:28-224): only when its weights are present
    Pp = E + "vit_pooler."
    try:
        te = get(Pp + "type_embed")
    except KeyError:
        te = None
    if te is not None:
        Cv, g = v.b2v, v.grid
        pos = sine_pos_embed(g, Cv)[0].to(te.device)                      # [P, C]; pos_embed is a non-persistent buffer
        yield "v.pool.emb", torch.cat([pos + te[0], pos + te[1]], 0).contiguous(), RDX_W_TENSOR
        for i in range(v.pool_blocks):
            Bk, o = f"{Pp}blocks.{i}.", f"v.pool.{i}."
            yield o + "n1_g", get(Bk + "norm1.weight").view(1, -1), RDX_W_F32
            yield o + "n1_b", get(Bk + "norm1.bias").view(1, -1), RDX_W_F32
            yield o + "n2_g", get(Bk + "norm2.weight").view(1, -1), RDX_W_F32
            yield o + "n2_b", get(Bk + "norm2.bias").view(1, -1), RDX_W_F32
            yield o + "wqkv", torch.cat([get(Bk + f"attn.proj_{n}.weight") for n in "qkv"], 0).contiguous(), RDX_W_GEMM
            yield o + "wo", get(Bk + "attn.proj.weight"), RDX_W_GEMM
            yield o + "bo", get(Bk + "attn.proj.bias").view(1, -1), RDX_W_F32
            yield o + "w1", get(Bk + "mlp.fc1.weight"), RDX_W_GEMM
            yield o + "b1", get(Bk + "mlp.fc1.bias").view(1, -1), RDX_W_F32
            yield o + "w2", get(Bk + "mlp.fc2.weight"), RDX_W_GEMM
            yield o + "b2", get(Bk + "mlp.fc2.bias").view(1, -1), RDX_W_F32
        yield "v.pool.norm_g", get(Pp + "norm_post.weight").view(1, -1), RDX_W_F32
        yield "v.pool.norm_b", get(Pp + "norm_post.bias").view(1, -1), RDX_W_F32
        # projector conv-1 over the real [patch_x | diff_x] channels: BN folded, nothing constant to fold
        yield "v.proj1f.w", (w0 * scale[:, None]).contiguous(), RDX_W_GEMM
        yield "v.proj1f.b", (get(J + "1.bias") - get(J + "1.running_mean") * scale).view(1, -1), RDX_W_F32
    yield "v.proj2.w", get(J + "3.weight").reshape(v.proj, v.proj).contiguous(), RDX_W_GEMM
    yield "v.proj2.b", get(J + "3.bias").view(1, -1), RDX_W_F32
    if with_ln:
        yield "v.ln.g", get("ln_vision.weight").view(1, -1), RDX_W_F32
        yield "v.ln.b", get("ln_vision.bias").view(1, -1), RDX_W_F32


def classifier_items(get: Getter, v: VisionCfg, c) -> Iterator[Item]:
    """ChexpertClassifier state_dict (findings_classifier/chexpert_model.py:8-13) -> engine tensors: the BioViL-T trunk +
    projector under `biovil_encoder.`, then fc1 / fc2. fc1's input is x.view(B, -1) of [B, C, g/pool, g/pool] -- the
    engine's pooling kernel writes exactly that (c, h, w) order, so the weight is used as is."""
    yield from vision_items(get, v, prefix="biovil_encoder.", with_ln=False)
    yield "cls.fc1.w", get("fc1.weight").contiguous(), RDX_W_GEMM
    yield "cls.fc1.b", get("fc1.bias").view(1, -1), RDX_W_F32
    yield "cls.fc2.w", get("fc2.weight").contiguous(), RDX_W_GEMM
    yield "cls.fc2.b", get("fc2.bias").view(1, -1), RDX_W_F32


def sine_pos_embed(grid: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """SinePositionEmbedding(embedding_dim=dim//2, normalize=True) over an all-ones grid mask
    (biovil_t/transformer.py:63-65,:248-266) -> [1, grid*grid, dim], built with the same torch ops."""
    import math
    npf = dim // 2
    ones = torch.ones(1, grid, grid)
    y = ones.cumsum(1, dtype=torch.float32)
    x = ones.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).view(1, grid * grid, dim)


def qformer_items(get: Getter, q: QFormerCfg) -> Iterator[Item]:
    B = "Qformer.bert."
    H = q.hidden
    qt = get("query_tokens").reshape(q.n_query, H)
    qln = F.layer_norm(qt, (H,), get(B + "embeddings.LayerNorm.weight"), get(B + "embeddings.LayerNorm.bias"), q.ln_eps)
    yield "q.query_ln", qln.contiguous(), RDX_W_TENSOR
    kv_w, kv_b = [], []
    for l in range(q.layers):
        L, o = f"{B}encoder.layer.{l}.", f"q{l}."
        a = L + "attention."
        yield o + "self.wqkv", torch.cat([get(a + f"self.{n}.weight") for n in ("query", "key", "value")], 0).contiguous(), RDX_W_GEMM
        yield o + "self.bqkv", torch.cat([get(a + f"self.{n}.bias") for n in ("query", "key", "value")], 0).view(1, -1), RDX_W_F32
        yield o + "self.wo", get(a + "output.dense.weight"), RDX_W_GEMM
        yield o + "self.bo", get(a + "output.dense.bias").view(1, -1), RDX_W_F32
        yield o + "self.ln_g", get(a + "output.LayerNorm.weight").view(1, -1), RDX_W_F32
        yield o + "self.ln_b", get(a + "output.LayerNorm.bias").view(1, -1), RDX_W_F32
        if q.has_cross(l):
            c = L + "crossattention."
            yield o + "cross.wq", get(c + "self.query.weight"), RDX_W_GEMM
            yield o + "cross.bq", get(c + "self.query.bias").view(1, -1), RDX_W_F32
            kv_w += [get(c + "self.key.weight"), get(c + "self.value.weight")]
            kv_b += [get(c + "self.key.bias"), get(c + "self.value.bias")]
            yield o + "cross.wo", get(c + "output.dense.weight"), RDX_W_GEMM
            yield o + "cross.bo", get(c + "output.dense.bias").view(1, -1), RDX_W_F32
            yield o + "cross.ln_g", get(c + "output.LayerNorm.weight").view(1, -1), RDX_W_F32
            yield o + "cross.ln_b", get(c + "output.LayerNorm.bias").view(1, -1), RDX_W_F32
        yield o + "ffn.w1", get(L + "intermediate_query.dense.weight"), RDX_W_GEMM
        yield o + "ffn.b1", get(L + "intermediate_query.dense.bias").view(1, -1), RDX_W_F32
        yield o + "ffn.w2", get(L + "output_query.dense.weight"), RDX_W_GEMM
        yield o + "ffn.b2", get(L + "output_query.dense.bias").view(1, -1), RDX_W_F32
        yield o + "ffn.ln_g", get(L + "output_query.LayerNorm.weight").view(1, -1), RDX_W_F32
        yield o + "ffn.ln_b", get(L + "output_query.LayerNorm.bias").view(1, -1), RDX_W_F32
    yield "q.cross.wkv", torch.cat(kv_w, 0).contiguous(), RDX_W_GEMM
    yield "q.cross.bkv", torch.cat(kv_b, 0).view(1, -1), RDX_W_F32


def rope_tables_f32(head_dim: int, max_pos: int, base: float):
    """cos/sin [max_pos, head_dim] in fp32, the torch op sequence of LlamaRotaryEmbedding.__init__ (:99-109)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def llama_items(get: Getter, c: LlamaCfg, lora: bool, fp8: bool = False, dtype=None) -> Iterator[Item]:
    # fp8: the decoder's GEMM weights (QKV + LoRA-A rows, o_proj, gate/up, down, lm_head) are additionally quantised to e4m3
    # with one scale per row (RDX_W_GEMM_FP8, BASELINE configs[4]); embeddings, norms, LoRA-B, img_proj keep the model dtype
    GK = RDX_W_GEMM_FP8 if fp8 else RDX_W_GEMM
    H, I = c.hidden, c.inter
    yield "embed", get("model.embed_tokens.weight"), RDX_W_TENSOR
    yield "final_norm", get("model.norm.weight").view(1, -1), RDX_W_TENSOR
    yield "lm_head", get("lm_head.weight"), GK
    yield "img_proj.w", get("model.img_proj_layer.weight"), RDX_W_GEMM
    # the decoder runs in the model dtype throughout (`.half()`, demo.py:234): img_proj_layer's bias is a half tensor in the reference and
    # F.linear adds THAT to the fp32 sum. The engine keeps biases as fp32 words, so round it here (an unrounded bias moves ~1 % of
    # the 32 spliced rows by one ulp, which the QKV projection then spreads over a few per cent of the K / V cache).
    b = get("model.img_proj_layer.bias")
    yield "img_proj.b", (b if dtype is None else b.to(dtype).float()).view(1, -1), RDX_W_F32
    cos, sin = rope_tables_f32(c.head_dim, c.max_pos, c.rope_base)
    yield "rope.cos", cos.contiguous(), RDX_W_TENSOR
    yield "rope.sin", sin.contiguous(), RDX_W_TENSOR
    for l in range(c.layers):
        L, o = f"model.layers.{l}.", f"l{l}."
        yield o + "attn_norm", get(L + "input_layernorm.weight").view(1, -1), RDX_W_TENSOR
        yield o + "mlp_norm", get(L + "post_attention_layernorm.weight").view(1, -1), RDX_W_TENSOR
        parts = [get(L + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")]
        if lora:
            parts += [get(L + "self_attn.q_proj.lora_A.weight"), get(L + "self_attn.v_proj.lora_A.weight")]
        yield o + "wqkv", torch.cat(parts, 0).contiguous(), GK
        del parts
        if lora:
            yield o + "lora_bq", get(L + "self_attn.q_proj.lora_B.weight").contiguous(), RDX_W_TENSOR
            yield o + "lora_bv", get(L + "self_attn.v_proj.lora_B.weight").contiguous(), RDX_W_TENSOR
        yield o + "wo", get(L + "self_attn.o_proj.weight"), GK
        g = get(L + "mlp.gate_proj.weight").view(I // 8, 1, 8, H)
        u = get(L + "mlp.up_proj.weight").view(I // 8, 1, 8, H)
        yield o + "wgu", torch.cat([g, u], 1).reshape(2 * I, H).contiguous(), GK
        del g, u
        yield o + "wdown", get(L + "mlp.down_proj.weight"), GK

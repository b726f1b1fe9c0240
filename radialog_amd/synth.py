"""Deterministic synthetic weights and inputs for the RaDialog hot path.

There is no network for checkpoints, so benchmarks and parity tests run on random-init weights of the
reference architecture (BASELINE.json configs). The generator is counter-based (a 32-bit integer hash
of the element index, seeded by the tensor NAME) and uses only exact integer / single-rounding float
operations, so the CPU box and the GPU box regenerate identical bytes, on CPU tensors and on device
tensors alike. `torch.manual_seed` is deliberately not used.

Tensor names are the reference `state_dict` keys:
  * vision   : `visual_encoder.encoder.encoder.*` (torchvision ResNet-50 keys, biovil_t/resnet.py:15-47),
               `visual_encoder.encoder.backbone_to_vit.weight`, `.missing_previous_emb` (biovil_t/encoder.py:102-108),
               `visual_encoder.projector.model.{0,1,3}.*` (biovil_t/modules.py:43-47), `ln_vision.*` (blip2.py:86)
  * Q-Former : `query_tokens`, `Qformer.bert.*` (Qformer.py:51-108,:111-400)
  * Llama    : `model.*`, `lm_head.weight` (modeling_llama_imgemb.py:441-448,:675-683),
               `model.img_proj_layer.*` (demo.py:229), LoRA `...{q_proj,v_proj}.lora_{A,B}.weight` (finetune.py:167-173)
"""
from __future__ import annotations

import math
import zlib
from typing import Callable, Dict, Iterator, List, Tuple

import torch

from .config import LlamaCfg, QFormerCfg, RaDialogCfg, VisionCfg, IMG_TOKEN_ID, N_IMG_TOKENS

_M32 = 0xFFFFFFFF
_CHUNK = 1 << 24


def _hash32(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """32-bit avalanche hash of int64 indices; every intermediate stays below 2^63."""
    x = (idx ^ seed) & _M32
    x = (x * 0x45D9F3B) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x2C1B3C6D) & _M32
    x = x ^ (x >> 15)
    return x


def uniform01(n: int, seed: int, device) -> torch.Tensor:
    """n floats in [0,1) with 24 random bits each (exactly representable in fp32)."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    for s in range(0, n, _CHUNK):
        e = min(n, s + _CHUNK)
        idx = torch.arange(s, e, dtype=torch.int64, device=device)
        h = _hash32(idx, seed)
        out[s:e] = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
    return out


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode()) & _M32


def synth(name: str, shape, lo: float, hi: float, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """Uniform[lo,hi) tensor determined by (name, shape)."""
    n = 1
    for s in shape:
        n *= int(s)
    u = uniform01(n, name_seed(name), device)
    t = (u * (hi - lo) + lo).reshape(tuple(shape))
    return t.to(dtype)


def _sym(name, shape, std, device, dtype=torch.float32):
    a = std * math.sqrt(3.0)          # uniform(-a, a) has standard deviation `std`
    return synth(name, shape, -a, a, device, dtype)


# ----------------------------------------------------------------------------------------------------
# weight specs: name -> (shape, generator)
# ----------------------------------------------------------------------------------------------------
Spec = Tuple[Tuple[int, ...], Callable[[str, Tuple[int, ...], object], torch.Tensor]]


def _w(std):
    return lambda name, shape, dev: _sym(name, shape, std, dev)


def _u(lo, hi):
    return lambda name, shape, dev: synth(name, shape, lo, hi, dev)


def vision_specs(v: VisionCfg, prefix: str = "visual_encoder.", with_ln: bool = True) -> Dict[str, Spec]:
    sp: Dict[str, Spec] = {}
    P = prefix + "encoder.encoder."

    def conv(name, cout, cin, k):
        sp[name] = ((cout, cin, k, k), _w(math.sqrt(2.0 / (cin * k * k))))

    def bn(name, c, gamma=(0.8, 1.2)):
        sp[name + ".weight"] = ((c,), _u(*gamma))
        sp[name + ".bias"] = ((c,), _u(-0.1, 0.1))
        sp[name + ".running_mean"] = ((c,), _u(-0.1, 0.1))
        sp[name + ".running_var"] = ((c,), _u(0.6, 1.4))

    conv(P + "conv1.weight", v.stem, 3, 7)
    bn(P + "bn1", v.stem)
    cin = v.stem
    for li, (planes, nblk) in enumerate(zip(v.planes, v.blocks), start=1):
        for b in range(nblk):
            pre = f"{P}layer{li}.{b}."
            conv(pre + "conv1.weight", planes, cin, 1)
            bn(pre + "bn1", planes)
            conv(pre + "conv2.weight", planes, planes, 3)
            bn(pre + "bn2", planes)
            conv(pre + "conv3.weight", planes * 4, planes, 1)
            bn(pre + "bn3", planes * 4, gamma=(0.2, 0.4))     # keeps the residual stream bounded
            if b == 0:
                conv(pre + "downsample.0.weight", planes * 4, cin, 1)
                bn(pre + "downsample.1", planes * 4, gamma=(0.5, 0.9))
            cin = planes * 4
    E = prefix + "encoder."
    sp[E + "backbone_to_vit.weight"] = ((v.b2v, v.trunk_out, 1, 1), _w(math.sqrt(1.0 / v.trunk_out)))
    sp[E + "missing_previous_emb"] = ((1, v.b2v, 1, 1), _w(0.5))
    Pp = E + "vit_pooler."                                   # two-image mode (biovil_t/transformer.py:28-65,:137-224)
    C = v.b2v
    for i in range(v.pool_blocks):
        Bk = f"{Pp}blocks.{i}."
        for nm in ("norm1", "norm2"):
            sp[Bk + nm + ".weight"] = ((C,), _u(0.8, 1.2))
            sp[Bk + nm + ".bias"] = ((C,), _u(-0.1, 0.1))
        for nm in ("proj_q", "proj_k", "proj_v"):
            sp[Bk + f"attn.{nm}.weight"] = ((C, C), _w(math.sqrt(1.0 / C)))
        sp[Bk + "attn.proj.weight"] = ((C, C), _w(math.sqrt(1.0 / C)))
        sp[Bk + "attn.proj.bias"] = ((C,), _u(-0.05, 0.05))
        for nm in ("fc1", "fc2"):
            sp[Bk + f"mlp.{nm}.weight"] = ((C, C), _w(math.sqrt(1.0 / C)))
            sp[Bk + f"mlp.{nm}.bias"] = ((C,), _u(-0.05, 0.05))
    sp[Pp + "norm_post.weight"] = ((C,), _u(0.8, 1.2))
    sp[Pp + "norm_post.bias"] = ((C,), _u(-0.1, 0.1))
    sp[Pp + "type_embed"] = ((2, 1, C), _w(0.2))
    J = prefix + "projector.model."
    sp[J + "0.weight"] = ((v.proj, 2 * v.b2v, 1, 1), _w(math.sqrt(2.0 / (2 * v.b2v))))
    bn(J + "1", v.proj)
    sp[J + "3.weight"] = ((v.proj, v.proj, 1, 1), _w(math.sqrt(1.0 / v.proj)))
    sp[J + "3.bias"] = ((v.proj,), _u(-0.05, 0.05))
    if with_ln:
        sp["ln_vision.weight"] = ((v.proj,), _u(0.8, 1.2))
        sp["ln_vision.bias"] = ((v.proj,), _u(-0.1, 0.1))
    return sp


def classifier_specs(v: VisionCfg, c) -> Dict[str, Spec]:
    """ChexpertClassifier (findings_classifier/chexpert_model.py:8-13): ImageModel under `biovil_encoder.` + fc1, fc2."""
    sp = vision_specs(v, prefix="biovil_encoder.", with_ln=False)
    gp = v.grid // c.pool
    feat = v.proj * gp * gp
    sp["fc1.weight"] = ((c.hidden, feat), _w(math.sqrt(2.0 / feat)))
    sp["fc1.bias"] = ((c.hidden,), _u(-0.1, 0.1))
    sp["fc2.weight"] = ((c.classes, c.hidden), _w(math.sqrt(1.0 / c.hidden)))
    sp["fc2.bias"] = ((c.classes,), _u(-0.5, 0.5))
    return sp


def qformer_specs(q: QFormerCfg) -> Dict[str, Spec]:
    sp: Dict[str, Spec] = {}
    H, I, W = q.hidden, q.inter, q.enc_width

    def lin(name, out, inp, std=0.04):
        sp[name + ".weight"] = ((out, inp), _w(std))
        sp[name + ".bias"] = ((out,), _u(-0.05, 0.05))

    def ln(name):
        sp[name + ".weight"] = ((H,), _u(0.8, 1.2))
        sp[name + ".bias"] = ((H,), _u(-0.1, 0.1))

    sp["query_tokens"] = ((1, q.n_query, H), _w(0.02))
    ln("Qformer.bert.embeddings.LayerNorm")
    for l in range(q.layers):
        L = f"Qformer.bert.encoder.layer.{l}."
        for nm in ("query", "key", "value"):
            lin(L + "attention.self." + nm, H, H)
        lin(L + "attention.output.dense", H, H)
        ln(L + "attention.output.LayerNorm")
        if q.has_cross(l):
            lin(L + "crossattention.self.query", H, H)
            lin(L + "crossattention.self.key", H, W)
            lin(L + "crossattention.self.value", H, W)
            lin(L + "crossattention.output.dense", H, H)
            ln(L + "crossattention.output.LayerNorm")
        lin(L + "intermediate_query.dense", I, H)
        lin(L + "output_query.dense", H, I, std=0.02)
        ln(L + "output_query.LayerNorm")
    return sp


N_PLANTED = 64           # lm_head rows with a decisive norm (see _lm_head)
PLANTED_STD = 1.5        # their logit standard deviation (x rms of the final hidden state ~ 1)
BACKGROUND_STD = 0.25    # logit standard deviation of every other row


def planted_rows(vocab: int) -> List[int]:
    """Token ids of the planted lm_head rows: spread over the vocabulary, never a special id (< 3) or <IMG>."""
    hi = min(vocab, IMG_TOKEN_ID) - 3
    return sorted({(1000 + 977 * k) % hi + 3 for k in range(N_PLANTED)})


def _lm_head(H: int):
    """Random-init logits of a 32k vocabulary have top-1/top-2 gaps of a few 1e-2: the fp16-vs-fp32 accumulation order
    flips the argmax and a token-identity test then asserts nothing ('hard parts'). So the head is drawn at
    two scales: 64 planted rows whose logits have standard deviation 1.5 and a background of 0.25. The greedy choice is the
    largest of 64 Gaussians that depend on the WHOLE hidden state (every kernel of the path still moves it), the typical
    top-2 gap is ~0.4 = hundreds of fp16 ulps, and |logit| stays below 8 where the fp16 ulp is under the 1e-2 tolerance."""
    def gen(name, shape, dev):
        t = _sym(name, shape, BACKGROUND_STD / math.sqrt(H), dev)
        rows = torch.tensor(planted_rows(shape[0]), dtype=torch.int64, device=t.device)
        t[rows] = t[rows] * (PLANTED_STD / BACKGROUND_STD)
        return t
    return gen


def llama_specs(c: LlamaCfg, lora: bool = True) -> Dict[str, Spec]:
    sp: Dict[str, Spec] = {}
    H, I = c.hidden, c.inter
    sp["model.embed_tokens.weight"] = ((c.vocab, H), _w(0.02))
    for l in range(c.layers):
        L = f"model.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sp[L + f"self_attn.{nm}.weight"] = ((H, H), _w(0.02))
        sp[L + "mlp.gate_proj.weight"] = ((I, H), _w(0.02))
        sp[L + "mlp.up_proj.weight"] = ((I, H), _w(0.02))
        sp[L + "mlp.down_proj.weight"] = ((H, I), _w(0.02))
        sp[L + "input_layernorm.weight"] = ((H,), _u(0.8, 1.2))
        sp[L + "post_attention_layernorm.weight"] = ((H,), _u(0.8, 1.2))
        if lora:
            for nm in ("q_proj", "v_proj"):
                sp[L + f"self_attn.{nm}.lora_A.weight"] = ((c.lora_r, H), _w(0.02))
                sp[L + f"self_attn.{nm}.lora_B.weight"] = ((H, c.lora_r), _w(0.02))
    sp["model.norm.weight"] = ((H,), _u(0.8, 1.2))
    sp["lm_head.weight"] = ((c.vocab, H), _lm_head(H))
    sp["model.img_proj_layer.weight"] = ((H, c.qformer_dim), _w(0.02))
    sp["model.img_proj_layer.bias"] = ((H,), _u(-0.02, 0.02))
    return sp


def iter_weights(specs: Dict[str, Spec], device="cpu") -> Iterator[Tuple[str, torch.Tensor]]:
    """Generate fp32 tensors one at a time (the full Vicuna-7B set is 26 GB in fp32)."""
    for name, (shape, gen) in specs.items():
        yield name, gen(name, shape, device)


def make_weights(specs: Dict[str, Spec], device="cpu", dtype=None) -> Dict[str, torch.Tensor]:
    out = {}
    for name, t in iter_weights(specs, device):
        out[name] = t if dtype is None else t.to(dtype)
    return out


# ----------------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------------
def synth_images(batch: int, size: int = 448, device="cpu", seed: int = 16) -> torch.Tensor:
    """`float32[B,3,size,size]` in [0,1]: u8/255 values, three identical channels (the reference's
    ExpandChannels, model/lavis/data/ReportDataset.py:80-93), low-pass filtered so it looks like a
    radiograph. Content is irrelevant to timing."""
    coarse = size // 8
    u = uniform01(batch * coarse * coarse, seed, device).reshape(batch, 1, coarse, coarse)
    img = torch.nn.functional.interpolate(u, size=(size, size), mode="nearest")
    fine = uniform01(batch * size * size, seed + 1, device).reshape(batch, 1, size, size)
    img = torch.floor((0.8 * img + 0.2 * fine) * 255.0) / 255.0
    return img.repeat(1, 3, 1, 1).contiguous()


def synth_prompt_ids(batch: int, length: int = 160, vocab: int = 32001, img_offset: int = 20,
                     pad_rows: bool = False, device="cpu", seed: int = 7) -> torch.Tensor:
    """`int64[B,length]`: BOS(1) + ids in [3,31999] with 32 x `<IMG>`(32000) at `img_offset` from the first
    real token; with `pad_rows`, every 4th row is left-padded with pad id 0 (lengths 120..length), like
    `tokenizer.batch_encode_plus(padding=True)` with `padding_side='left'` (test.py:291,:336)."""
    hi = min(vocab, IMG_TOKEN_ID) - 3
    u = uniform01(batch * length, seed, device).reshape(batch, length)
    ids = (u * hi).to(torch.int64) + 3
    for b in range(batch):
        pad = 0
        if pad_rows and b % 4 == 3:
            pad = (b * 7) % 41
        if pad:
            ids[b, :pad] = 0
        ids[b, pad] = 1
        ids[b, pad + img_offset: pad + img_offset + N_IMG_TOKENS] = IMG_TOKEN_ID
    return ids

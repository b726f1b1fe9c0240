"""Shape configuration of the RaDialog hot path (image encode -> prompt project -> Llama greedy decode).

The numbers restate what the reference instantiates, not code from it:
  * Vicuna-7B dims (lmsys/vicuna-7b-v1.3, loaded at /root/reference/demo.py:224-225, test.py:291-292);
    `<IMG>` token id 32000 (modeling_llama_imgemb.py:500), vocab 32000 (demo.py) or 32001 after
    `resize_token_embeddings` (test.py:297); LoRA r=8 alpha=16 on q_proj,v_proj (finetune.py:167-173).
  * Q-Former: bert-base dims + cross-attention every 2nd layer to 1408-wide image tokens, 32 queries
    (model/lavis/models/blip2_models/blip2.py:47-62).
  * BioViL-T: torchvision ResNet-50 trunk (biovil_t/resnet.py:73-80), 1x1 backbone_to_vit 2048->256
    (biovil_t/encoder.py:102-103), projector MLP 512->1408->1408 (biovil_t/model.py:47-48), 14x14 grid.
"""
from dataclasses import dataclass, field
from typing import Tuple

IMG_TOKEN_ID = 32000      # modeling_llama_imgemb.py:500
N_IMG_TOKENS = 32         # modeling_llama_imgemb.py:502-517


@dataclass(frozen=True)
class LlamaCfg:
    vocab: int = 32001
    hidden: int = 4096
    inter: int = 11008
    layers: int = 32
    heads: int = 32
    rope_base: float = 10000.0
    max_pos: int = 2048
    rms_eps: float = 1e-6
    lora_r: int = 8
    lora_alpha: int = 16
    qformer_dim: int = 768

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def lora_scale(self) -> float:
        return self.lora_alpha / self.lora_r


@dataclass(frozen=True)
class QFormerCfg:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    enc_width: int = 1408
    n_query: int = 32
    cross_freq: int = 2
    ln_eps: float = 1e-12

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def has_cross(self, layer: int) -> bool:
        return layer % self.cross_freq == 0


@dataclass(frozen=True)
class VisionCfg:
    img: int = 448
    stem: int = 64                                  # conv1 output channels
    planes: Tuple[int, ...] = (64, 128, 256, 512)   # Bottleneck planes (x4 expansion)
    blocks: Tuple[int, ...] = (3, 4, 6, 3)
    b2v: int = 256                                  # backbone_to_vit output channels
    proj: int = 1408                                # projector hidden = output
    bn_eps: float = 1e-5
    ln_eps: float = 1e-5                            # ln_vision (nn.LayerNorm default)
    pool_blocks: int = 3                            # VisionTransformerPooler (biovil_t/transformer.py:42-52): two-image mode
    pool_heads: int = 8
    pool_ln_eps: float = 1e-6

    @property
    def grid(self) -> int:
        """Side of the trunk's output grid: conv1 /2, maxpool /2, then three stride-2 stages of (g - 1) // 2 + 1
        (448 -> 14; 488, the findings classifier's crop, -> 16)."""
        g = self.img // 4
        for _ in range(3):
            g = (g - 1) // 2 + 1
        return g

    @property
    def trunk_out(self) -> int:
        return self.planes[-1] * 4

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid


@dataclass(frozen=True)
class ClsCfg:
    """ChexpertClassifier head (findings_classifier/chexpert_model.py:7-21): avg_pool2d(pool) over the projected patch grid,
    fc1 -> ReLU -> fc2; 14 CheXpert classes (demo.py:157-163)."""
    hidden: int = 512
    classes: int = 14
    pool: int = 4


@dataclass(frozen=True)
class RaDialogCfg:
    llama: LlamaCfg = field(default_factory=LlamaCfg)
    qformer: QFormerCfg = field(default_factory=QFormerCfg)
    vision: VisionCfg = field(default_factory=VisionCfg)
    cls: ClsCfg = field(default_factory=ClsCfg)


def full_cfg(vocab: int = 32001) -> RaDialogCfg:
    """The shapes BASELINE.json's configs are quoted on."""
    return RaDialogCfg(llama=LlamaCfg(vocab=vocab))


def classifier_cfg() -> RaDialogCfg:
    """The findings classifier as demo.py runs it: BioViL-T with its stock 128-wide projector (get_biovil_t_image_encoder,
    joint_feature_size 128) on a 488 px centre crop -> 16x16 grid -> avg_pool2d(4) -> 128*4*4 -> 512 -> 14."""
    return RaDialogCfg(vision=VisionCfg(img=488, proj=128), cls=ClsCfg())


def small_classifier_cfg() -> RaDialogCfg:
    """Reduced classifier for parity tests; 136 px exercises the odd intermediate grid sizes (34 -> 17 -> 9 -> 5)."""
    return RaDialogCfg(vision=VisionCfg(img=136, stem=32, planes=(32, 64, 128, 256), blocks=(1, 2, 2, 1), b2v=64, proj=64),
                       cls=ClsCfg(hidden=96, classes=14, pool=2))


def small_cfg() -> RaDialogCfg:
    """Reduced shapes for parity tests the CPU oracle finishes in seconds. Head dims stay at the
    real 128 (Llama) / 64 (Q-Former) so the same kernel instantiations are exercised."""
    return RaDialogCfg(
        llama=LlamaCfg(vocab=32001, hidden=512, inter=1408, layers=2, heads=4, qformer_dim=192),
        qformer=QFormerCfg(hidden=192, layers=4, heads=3, inter=768, enc_width=352, n_query=32),
        vision=VisionCfg(img=128, stem=32, planes=(32, 64, 128, 256), blocks=(1, 2, 2, 1), b2v=64, proj=352,
                         pool_blocks=2, pool_heads=2),
    )

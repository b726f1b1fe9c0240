"""SURVEY.md 8f rank 2 -- the on-disk weight formats of the released RaDialog artefacts, loaded through the public surface:

  * peft adapter directory (finetune.py:121-150): `adapter_model.bin` with peft key names + `img_proj_layer`, `adapter_config.json`
  * LAVIS `checkpoint_N.pth` (runner_base.py:658-683): {'model': state_dict WITHOUT the frozen parameters, ...}
  * BioViL-T `biovil_t_image_model_proj_size_128.pt` (biovil_t/model.py:56-65): `projector.*` keys dropped at load
  * missing files / hub ids raise instead of silently running on random weights (base_model.py:43-44)

The files are synthesised here from the deterministic generator under the reference's key names. CPU tests cover the name
mapping and the error behaviour; the `gpu` tests load them into the engine and compare with the directly-fed weights."""
import json
import os
import types
import warnings

import pytest
import torch

from radialog_amd import synth
from radialog_amd.config import small_cfg


# ---------------------------------------------------------------------------------------------------- file builders
def write_adapter_dir(path, lcfg, W, r=8, alpha=16, targets=("q_proj", "v_proj"), default_infix=True):
    """What finetune.py's save_pretrained leaves behind: peft's get_peft_model_state_dict keys + the img_proj_layer pair."""
    os.makedirs(path, exist_ok=True)
    sd = {}
    for l in range(lcfg.layers):
        for m in ("q_proj", "v_proj"):
            for ab in "AB":
                k = f"model.layers.{l}.self_attn.{m}.lora_{ab}.weight"
                pk = "base_model.model." + (k.replace(f".lora_{ab}.", f".lora_{ab}.default.") if default_infix else k)
                sd[pk] = W[k].half()
    sd["base_model.model.model.img_proj_layer.weight"] = W["model.img_proj_layer.weight"].half()
    sd["base_model.model.model.img_proj_layer.bias"] = W["model.img_proj_layer.bias"].half()
    torch.save(sd, os.path.join(path, "adapter_model.bin"))
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump({"base_model_name_or_path": "lmsys/vicuna-7b-v1.3", "bias": "none", "fan_in_fan_out": False, "inference_mode": True,
                   "lora_alpha": alpha, "lora_dropout": 0.05, "peft_type": "LORA", "r": r, "target_modules": list(targets),
                   "task_type": "CAUSAL_LM"}, f)
    return path


def write_hf_dir(path, lcfg, W):
    """A local Vicuna checkpoint directory: base weights only (no img_proj_layer, no LoRA), fp16 safetensors."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    base = {k: v.half().contiguous() for k, v in W.items() if "lora_" not in k and "img_proj_layer" not in k}
    save_file(base, os.path.join(path, "model.safetensors"))
    return path


def write_biovil_t_pt(path, vcfg, W):
    """ImageModel state dict of the BioViL-T release: trunk + pooler under `encoder.`, and the STOCK 128-wide projector."""
    sd = {}
    for k, v in W.items():
        if k.startswith("visual_encoder.encoder."):
            sd[k[len("visual_encoder."):]] = v.clone()
            if k.endswith("running_var"):
                sd[k[len("visual_encoder."):].replace("running_var", "num_batches_tracked")] = torch.tensor(1000)
    sd["projector.model.0.weight"] = torch.randn(128, 2 * vcfg.b2v, 1, 1)
    sd["projector.model.1.weight"] = torch.ones(128)
    sd["projector.model.1.bias"] = torch.zeros(128)
    sd["projector.model.1.running_mean"] = torch.zeros(128)
    sd["projector.model.1.running_var"] = torch.ones(128)
    sd["projector.model.3.weight"] = torch.randn(128, 128, 1, 1)
    sd["projector.model.3.bias"] = torch.zeros(128)
    torch.save(sd, path)
    return path


def write_lavis_checkpoint(path, cfg, W, with_projector=False):
    """runner_base.py:658-683: parameters with requires_grad=False (the whole visual_encoder: conv, BN affine, projector) are
    deleted from the state dict before saving; BUFFERS (BatchNorm running statistics) and everything trainable stay."""
    model = {}
    for k, v in W.items():
        if k.startswith("visual_encoder."):
            is_buffer = k.endswith("running_mean") or k.endswith("running_var")
            if is_buffer or (with_projector and ".projector." in k):
                model[k] = v.clone()
        else:
            model[k] = v.clone()                                         # query_tokens, ln_vision.*, Qformer.bert.*
    H = cfg.qformer.hidden
    model["vision_proj.weight"] = torch.randn(256, H)                    # trained heads that the inference path never touches
    model["text_proj.weight"] = torch.randn(256, H)
    model["itm_head.weight"] = torch.randn(2, H)
    model["temp"] = torch.tensor(0.07)
    model["Qformer.bert.embeddings.position_ids"] = torch.arange(512)[None]
    torch.save({"model": model, "optimizer": {"state": {}}, "config": {"run": {}}, "scaler": None, "epoch": 4}, path)
    return path


@pytest.fixture(scope="module")
def cfg():
    return small_cfg()


@pytest.fixture(scope="module")
def W(cfg):
    return synth.make_weights({**synth.vision_specs(cfg.vision), **synth.qformer_specs(cfg.qformer), **synth.llama_specs(cfg.llama)})


def _blip(cfg, **kw):
    from radialog_amd.blip2_qformer import Blip2Qformer
    return Blip2Qformer(img_size=cfg.vision.img, dtype="f16", cfg=cfg, **kw)


# ---------------------------------------------------------------------------------------------------- CPU: mapping + errors
def test_no_silent_random_weights(tmp_path):
    from radialog_amd.blip2_qformer import Blip2Qformer, Config, tasks
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM
    with pytest.raises(OSError):
        LlamaForCausalLM.from_pretrained("lmsys/vicuna-7b-v1.3", torch_dtype=torch.float16, device_map="auto")     # demo.py:225 verbatim
    with pytest.raises(OSError):
        LlamaForCausalLM.from_pretrained(str(tmp_path / "typo"), torch_dtype=torch.float16)
    with pytest.raises(OSError):
        LlamaForCausalLM.from_pretrained(None)
    p = tmp_path / "c.yaml"
    p.write_text("model:\n  arch: blip2\n  vit_model: biovil\n  image_size: 448\n  num_query_token: 32\n  load_finetuned: true\n"
                 f"  finetuned: {tmp_path}/nope.pth\n  biovil_t_weights: {tmp_path}/nope.pt\n")
    cfg = Config(types.SimpleNamespace(cfg_path=str(p), options=None))
    with pytest.raises(RuntimeError):                                     # base_model.py:43-44 "checkpoint url or path is invalid"
        tasks.setup_task(cfg).build_model(cfg)
    with pytest.raises(RuntimeError, match="no CPU implementation|no weights"):
        Blip2Qformer().forward_image(torch.zeros(1, 3, 448, 448))


def test_adapter_directory_is_read_with_its_config(tmp_path, cfg, W):
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM, PeftModelForCausalLM
    lm = LlamaForCausalLM.from_pretrained(write_hf_dir(str(tmp_path / "vicuna"), cfg.llama, W), torch_dtype=torch.float16, cfg=cfg.llama)
    assert lm.lora is False and "model.img_proj_layer.weight" not in lm._state
    peft = PeftModelForCausalLM.from_pretrained(lm, write_adapter_dir(str(tmp_path / "ad"), cfg.llama, W), torch_dtype=torch.float16,
                                                use_ram_optimized_load=False).half().eval()
    assert peft.base_model.model is lm and lm.lora is True and lm.lcfg.lora_r == 8 and lm.lcfg.lora_scale == 2.0
    for k in ("model.layers.1.self_attn.v_proj.lora_B.weight", "model.layers.0.self_attn.q_proj.lora_A.weight",
              "model.img_proj_layer.weight", "model.img_proj_layer.bias"):
        assert torch.equal(lm._state[k], W[k].half().float()), k
    # peft versions without the `.default.` adapter-name infix write the same keys minus it
    lm2 = LlamaForCausalLM.from_pretrained(str(tmp_path / "vicuna"), cfg=cfg.llama)
    lm2.load_adapter(write_adapter_dir(str(tmp_path / "ad2"), cfg.llama, W, default_infix=False))
    assert torch.equal(lm2._state["model.layers.0.self_attn.v_proj.lora_A.weight"], W["model.layers.0.self_attn.v_proj.lora_A.weight"].half().float())
    # alpha is taken from the file, not from the defaults
    lm3 = LlamaForCausalLM.from_pretrained(str(tmp_path / "vicuna"), cfg=cfg.llama)
    lm3.load_adapter(write_adapter_dir(str(tmp_path / "ad3"), cfg.llama, W, alpha=32))
    assert lm3.lcfg.lora_scale == 4.0
    # unsupported adapters fail at load, with the reason
    for kw, exc in (({"r": 16}, ValueError), ({"targets": ("q_proj", "k_proj", "v_proj", "o_proj")}, ValueError)):
        with pytest.raises(exc):
            LlamaForCausalLM.from_pretrained(str(tmp_path / "vicuna"), cfg=cfg.llama).load_adapter(
                write_adapter_dir(str(tmp_path / ("bad" + str(len(str(kw))))), cfg.llama, W, **kw))
    with pytest.raises(OSError):
        LlamaForCausalLM.from_pretrained(str(tmp_path / "vicuna"), cfg=cfg.llama).load_adapter(str(tmp_path / "vicuna"))


def test_lavis_checkpoint_and_biovil_t_file_assemble_the_encoder_state(tmp_path, cfg, W):
    pt = write_biovil_t_pt(str(tmp_path / "biovil_t_image_model_proj_size_128.pt"), cfg.vision, W)
    ck = write_lavis_checkpoint(str(tmp_path / "checkpoint_4.pth"), cfg, W)
    m = _blip(cfg).load_biovil_t(pt).load_checkpoint(ck)
    st = m._weights[1]
    assert not any(k.startswith("visual_encoder.projector.model.0") or k.startswith("projector") for k in st), "stock projector must be dropped"
    assert "visual_encoder.projector.model.1.running_mean" in st                    # buffers DO come with the LAVIS checkpoint
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.manual_seed(3)
        get = m._resolve_weights()
    assert any("projector" in str(w.message) for w in rec), "drawing the unsaved projector must be announced"
    for k in ("visual_encoder.encoder.encoder.layer2.1.conv2.weight", "visual_encoder.encoder.backbone_to_vit.weight",
              "Qformer.bert.encoder.layer.2.crossattention.self.key.weight", "query_tokens", "ln_vision.bias",
              "visual_encoder.encoder.vit_pooler.type_embed"):
        assert torch.equal(get(k), W[k]), k
    p0 = get("visual_encoder.projector.model.0.weight")
    assert p0.shape == (cfg.vision.proj, 2 * cfg.vision.b2v, 1, 1) and float(p0.abs().max()) <= 1.0 / (2 * cfg.vision.b2v) ** 0.5
    assert torch.equal(get("visual_encoder.projector.model.1.weight"), torch.ones(cfg.vision.proj))
    # a checkpoint that does carry the projector wins over the default draw, silently
    m2 = _blip(cfg).load_biovil_t(pt).load_checkpoint(write_lavis_checkpoint(str(tmp_path / "c2.pth"), cfg, W, with_projector=True))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        g2 = m2._resolve_weights()
    assert torch.equal(g2("visual_encoder.projector.model.3.weight"), W["visual_encoder.projector.model.3.weight"])
    # the LAVIS file alone is not enough: the frozen trunk is missing, and that is an error, not random weights
    with pytest.raises(RuntimeError, match="lacks"):
        _blip(cfg).load_checkpoint(ck)._resolve_weights()
    with pytest.raises(RuntimeError, match="invalid"):
        _blip(cfg).load_checkpoint(str(tmp_path / "missing.pth"))


# ---------------------------------------------------------------------------------------------------- GPU: the loaded files compute
@pytest.mark.gpu
def test_released_file_formats_drive_the_engine_like_direct_weights(tmp_path, cfg, W):
    """Files -> public surface -> engine must give exactly what the same tensors fed directly give (bitwise: same kernels,
    same bytes after the fp16 round trip of the adapter / safetensors files), and match the oracle on those weights."""
    from oracle import ref_cpu
    from radialog_amd.blip2_qformer import Config, tasks
    from radialog_amd.engine import RdxEngine
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM, PeftModelForCausalLM
    from _parity import check_greedy
    pt = write_biovil_t_pt(str(tmp_path / "biovil_t.pt"), cfg.vision, W)
    ck = write_lavis_checkpoint(str(tmp_path / "checkpoint_4.pth"), cfg, W, with_projector=True)
    y = tmp_path / "c.yaml"
    y.write_text(f"model:\n  arch: blip2\n  vit_model: biovil\n  image_size: {cfg.vision.img}\n  num_query_token: 32\n  dtype: f16\n"
                 f"  load_finetuned: true\n  finetuned: {ck}\n  biovil_t_weights: {pt}\n")
    conf = Config(types.SimpleNamespace(cfg_path=str(y), options=None))
    from radialog_amd import blip2_qformer as bq
    orig = bq.Blip2Qformer.__init__
    try:                                                # the YAML names the full-size model; shrink it to the test shapes
        bq.Blip2Qformer.__init__ = lambda self, **kw: orig(self, cfg=cfg, **kw)
        blip = tasks.setup_task(conf).build_model(conf).to(torch.device("cuda")).eval()
    finally:
        bq.Blip2Qformer.__init__ = orig
    img = synth.synth_images(2, cfg.vision.img)
    q, emb = blip.forward_image(img.cuda())
    eng = RdxEngine(cfg, dtype="f16", device=0, llama=False)
    eng.load_weights(lambda n: W[n].cuda(), llama=False)
    q2, emb2 = eng.encode_image(img.cuda())
    assert torch.equal(q, q2) and torch.equal(emb, emb2)
    eng.close()
    with torch.no_grad():
        rq, _ = ref_cpu.forward_image(img, W, cfg)
    assert float((q.cpu() - rq).norm() / rq.norm()) < 5e-3

    lm = LlamaForCausalLM.from_pretrained(write_hf_dir(str(tmp_path / "vicuna"), cfg.llama, W), torch_dtype=torch.float16,
                                          cfg=cfg.llama, max_batch=2, max_len=128)
    lm = PeftModelForCausalLM.from_pretrained(lm, write_adapter_dir(str(tmp_path / "ad"), cfg.llama, W), torch_dtype=torch.float16).half().eval()
    ids = synth.synth_prompt_ids(2, 48, vocab=cfg.llama.vocab, img_offset=4)
    out = lm.generate(input_ids=ids, qformer_embs=q, return_dict_in_generate=True, output_scores=True, max_new_tokens=8, eos_token_id=-1)
    Wh = {k: v.half().float() for k, v in W.items()}                                  # the files hold fp16
    with torch.no_grad():
        ref = ref_cpu.LlamaOracle(Wh, cfg.llama, torch.float16, lora=True).generate_greedy(ids, q.cpu(), max_new=8, eos_id=-1)
    check_greedy(out.sequences[:, 48:], torch.stack(out.scores), ref, 1e-2, 0.9, "adapter + safetensors files")

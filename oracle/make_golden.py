"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference). Imports the reference files by path (package
import dies on `omegaconf`, model/lavis/__init__.py:11), loads our deterministic synthetic weights into the
reference modules, runs them on CPU and stores small inputs/outputs as .npz. The weights themselves are NOT
stored -- tests regenerate them from `radialog_amd.synth`. The reference's source never enters this repo.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz

Pinned here:
  llama_*    modeling_llama_imgemb.py: LlamaForCausalLM.forward / prepare_inputs_for_generation driven by a
             hand greedy loop (HF 4.28.1 rule), image splice via the `dicom` side channel, left padding,
             the no-<IMG> quirk; fp32, fp16 and bf16.
  qformer_*  Qformer.py: BertLMHeadModel(...).bert(query_embeds, encoder_hidden_states, encoder_attention_mask)
  projector  biovil_t/modules.py MLP(use_1x1_convs=True) in eval mode (+ the reshape scramble + LayerNorm restated
             with torch ops exactly as blip2_qformer.py:469 writes them)
  rope/rms   LlamaRotaryEmbedding tables, LlamaRMSNorm rows
  downstream downstream_tasks/automated_correction.py + chexpert_classification_downstream.py prompt builders (strings)
  llama_real_layer   ONE LlamaDecoderLayer at the production shape (4096 / 11008 / 32 x 128): prefill T = 8 + 2 decode steps, three dtypes
  qformer_real       BertLMHeadModel(...).bert at the production shape (768 x 12, 32 queries x 196 x 1408), B = 1
  vit_pooler biovil_t/transformer.py VisionTransformerPooler (Block, MultiHeadAttentionLayer, SinePositionEmbedding are the
             reference's own code) in eval mode, two-image call. Its three timm==0.4.12 imports are shimmed in-process: DropPath
             (identity in eval), trunc_normal_ (init only, overwritten by our weights) and Mlp, restated as timm 0.4.12 defines
             it (fc1 -> act -> drop -> fc2 -> drop): the Mlp stays 'parity unpinned', everything else in the pooler is pinned.
"""
import importlib.util
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

from radialog_amd import synth                                   # noqa: E402
from radialog_amd.config import LlamaCfg, QFormerCfg, VisionCfg  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def golden_llama_cfg():
    return LlamaCfg(vocab=512, hidden=128, inter=256, layers=2, heads=4, qformer_dim=48)


def golden_qformer_cfg():
    return QFormerCfg(hidden=192, layers=4, heads=3, inter=384, enc_width=96, n_query=32)


def make_llama():
    import transformers
    from transformers import LlamaConfig
    c = golden_llama_cfg()
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pretraining", "embs"))
    qf = synth.synth("golden.qformer_embs", (3, 32, c.qformer_dim), -1.0, 1.0)
    dic = {f"dicom{i}": qf[i].numpy() for i in range(3)}
    with open(os.path.join(work, "pretraining/embs/stage1_pt_instruct_blip_origlr_img448_embeddings_test.pkl"), "wb") as f:
        pickle.dump(dic, f)
    cwd = os.getcwd()
    W = synth.make_weights(synth.llama_specs(c, lora=False))

    def build(dt):
        """A fresh reference model per dtype: `.to(dt)` rounds parameters AND the cached rope tables, so
        chaining casts on one instance would double-round them."""
        os.chdir(work)                         # the reference opens the pkl relative to CWD (:461)
        try:
            ref = _load("model/lavis/models/blip2_models/modeling_llama_imgemb.py", "ref_llama")
            hcfg = LlamaConfig(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.inter,
                               num_hidden_layers=c.layers, num_attention_heads=c.heads,
                               max_position_embeddings=c.max_pos, rms_norm_eps=c.rms_eps, pad_token_id=0,
                               hidden_act="silu")
            mdl = ref.LlamaForCausalLM(hcfg)
        finally:
            os.chdir(cwd)
        mdl.model.img_proj_layer = torch.nn.Linear(c.qformer_dim, c.hidden)      # demo.py:229
        missing, unexpected = mdl.load_state_dict(dict(W), strict=False)
        assert not unexpected, unexpected
        assert all("inv_freq" in k for k in missing), missing
        return mdl.eval().to(dt)

    T = 44
    ids = synth.synth_prompt_ids(3, T, vocab=c.vocab, img_offset=5, seed=11)
    ids[1, :6] = 0                                   # left padding on row 1 (pad id 0)
    ids[1, 6] = 1
    ids[1, 7:] = synth.synth_prompt_ids(1, T - 7, vocab=c.vocab, img_offset=4, seed=12)[0][: T - 7]
    ids[1, 7] = 5
    ids[2] = synth.synth_prompt_ids(1, T, vocab=c.vocab, img_offset=3, seed=13)[0]
    ids[2][ids[2] == 32000] = 77                     # row 2: no <IMG> -> the drop-32-tokens quirk
    dicom = ["dicom0", "dicom1", "dicom2"]
    out = {"ids": ids.numpy(), "qformer_embs": qf.numpy()}

    n_new = 8
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        m = build(dt)      # rotary tables follow .half() like every floating buffer (demo.py:234)
        with torch.no_grad():
            am = ids.ne(0).long()
            seq, past = ids.clone(), None
            toks, step_logits = [], []
            unfinished = torch.ones(ids.shape[0], dtype=torch.long)
            for step in range(n_new):
                mi = m.prepare_inputs_for_generation(seq, past_key_values=past, attention_mask=am,
                                                     use_cache=True, dicom=dicom)
                o = m(**mi, return_dict=True)
                if step == 0:
                    out[f"prefill_logits_{tag}"] = o.logits.float().numpy()
                    out[f"prefill_k0_{tag}"] = o.past_key_values[0][0].float().numpy()
                    out[f"prefill_v1_{tag}"] = o.past_key_values[1][1].float().numpy()
                past = o.past_key_values
                row = o.logits[:, -1, :]
                step_logits.append(row.float().numpy())
                nxt = row.argmax(-1)
                nxt = nxt * unfinished + 0 * (1 - unfinished)
                toks.append(nxt.numpy())
                seq = torch.cat([seq, nxt[:, None]], dim=-1)
                am = torch.cat([am, am.new_ones(am.shape[0], 1)], dim=-1)
                unfinished = unfinished * (nxt != 2).long()
            out[f"tokens_{tag}"] = np.stack(toks, 1)
            out[f"step_logits_{tag}"] = np.stack(step_logits, 0)
    # pieces
    m = build(torch.float32)
    rot = m.model.layers[0].self_attn.rotary_emb
    cos, sin = rot(torch.zeros(1, 1, 1, c.head_dim), seq_len=64)
    out["rope_cos_f32"] = cos[0, 0].numpy()
    out["rope_sin_f32"] = sin[0, 0].numpy()
    xr = synth.synth("golden.rms_in", (5, c.hidden), -3.0, 3.0)
    out["rms_in"] = xr.numpy()
    out["rms_out_f32"] = m.model.norm(xr).detach().numpy()
    out["rms_out_f16"] = build(torch.float16).model.norm(xr.half()).float().detach().numpy()
    # split_at_img quirks (:498-520)
    left, right = m.model.split_at_img(ids)
    out["split_left_len"] = np.array([len(t) for t in left])
    out["split_right_len"] = np.array([len(t) for t in right])
    np.savez_compressed(os.path.join(OUT, "llama_tiny.npz"), **out)
    print("llama_tiny:", {k: v.shape for k, v in out.items()})


def make_qformer():
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    # shims for transformers 5.x: names moved / removed since 4.28
    for nm in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, nm):
            setattr(mu, nm, getattr(pu, nm))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), torch.zeros(0))
    mu.PreTrainedModel.get_head_mask = lambda self, m, n, *a, **k: [None] * n
    ref = _load("model/lavis/models/blip2_models/Qformer.py", "ref_qformer")
    if hasattr(ref.BertPreTrainedModel, "init_weights"):
        ref.BertPreTrainedModel.init_weights = lambda self: None
    q = golden_qformer_cfg()
    bc = ref.BertConfig(vocab_size=64, hidden_size=q.hidden, num_hidden_layers=q.layers,
                        num_attention_heads=q.heads, intermediate_size=q.inter, max_position_embeddings=64,
                        layer_norm_eps=q.ln_eps, hidden_act="gelu", hidden_dropout_prob=0.1,
                        attention_probs_dropout_prob=0.1)
    bc.encoder_width = q.enc_width
    bc.add_cross_attention = True
    bc.cross_attention_freq = q.cross_freq
    bc.query_length = q.n_query
    model = ref.BertLMHeadModel(bc)
    W = synth.make_weights(synth.qformer_specs(q))
    sd = {k[len("Qformer."):]: v for k, v in W.items() if k.startswith("Qformer.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # everything on the query-only path must have been provided
    for mk in missing:
        assert any(s in mk for s in ("word_embeddings", "position_embeddings", "position_ids", ".intermediate.dense",
                                     ".output.dense", ".output.LayerNorm", "cls.")), mk
    model.eval()
    B, P = 2, 20
    img = synth.synth("golden.qf_img", (B, P, q.enc_width), -2.0, 2.0)
    with torch.no_grad():
        qt = W["query_tokens"].expand(B, -1, -1)
        o = model.bert(query_embeds=qt, encoder_hidden_states=img,
                       encoder_attention_mask=torch.ones(B, P, dtype=torch.long), return_dict=True)
    np.savez_compressed(os.path.join(OUT, "qformer_small.npz"), img=img.numpy(),
                        out=o.last_hidden_state.numpy())
    print("qformer_small:", o.last_hidden_state.shape, float(o.last_hidden_state.abs().mean()))


def _qformer_ref():
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for nm in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, nm):
            setattr(mu, nm, getattr(pu, nm))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), torch.zeros(0))
    mu.PreTrainedModel.get_head_mask = lambda self, m, n, *a, **k: [None] * n
    ref = _load("model/lavis/models/blip2_models/Qformer.py", "ref_qformer")
    if hasattr(ref.BertPreTrainedModel, "init_weights"):
        ref.BertPreTrainedModel.init_weights = lambda self: None
    return ref


def _u16(t):
    """fp16 / bf16 values as their 16 bits (halves the fixture)."""
    return t.contiguous().view(torch.int16).numpy()


REAL_LAYER_T, REAL_LAYER_PAD, REAL_LAYER_STEPS = 8, 3, 2
REAL_LAYER_HEADS = (0, 13, 31)           # K / V heads kept in the fixture (all of hidden_states is kept)


def real_layer_inputs(c):
    """Inputs of the real-shape decoder-layer fixture (regenerated by the test, not stored): B = 2, row 1 left-padded by 3."""
    T, S = REAL_LAYER_T, REAL_LAYER_STEPS
    xs = [synth.synth("golden.real_layer.x0", (2, T, c.hidden), -1.0, 1.0)]
    xs += [synth.synth(f"golden.real_layer.x{s + 1}", (2, 1, c.hidden), -1.0, 1.0) for s in range(S)]
    am = torch.ones(2, T, dtype=torch.long)
    am[1, :REAL_LAYER_PAD] = 0
    return xs, am


def make_llama_real_layer():
    """ONE real-shape LlamaDecoderLayer (hidden 4096, inter 11008, 32 heads x 128) of the reference
    (modeling_llama_imgemb.py:253-318, attention :187-250), prefill T = 8 + 2 decode steps, fp32 / fp16 / bf16, with the masks the
    reference's own _make_causal_mask / _expand_mask build (:44-73, :475-496) and prepare_inputs' position rule. Stores OUTPUTS only;
    inputs and weights are regenerated from radialog_amd.synth by the test."""
    from transformers import LlamaConfig
    c = LlamaCfg(layers=1)
    work = tempfile.mkdtemp()
    os.makedirs(os.path.join(work, "pretraining", "embs"))
    with open(os.path.join(work, "pretraining/embs/stage1_pt_instruct_blip_origlr_img448_embeddings_test.pkl"), "wb") as f:
        pickle.dump({}, f)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        ref = _load("model/lavis/models/blip2_models/modeling_llama_imgemb.py", "ref_llama_real")
    finally:
        os.chdir(cwd)
    hcfg = LlamaConfig(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.inter, num_hidden_layers=1,
                       num_attention_heads=c.heads, max_position_embeddings=c.max_pos, rms_norm_eps=c.rms_eps, pad_token_id=0, hidden_act="silu")
    specs = {k: v for k, v in synth.llama_specs(c, lora=False).items() if k.startswith("model.layers.0.")}
    W = synth.make_weights(specs)
    xs, am = real_layer_inputs(c)
    out = {}
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        layer = ref.LlamaDecoderLayer(hcfg)
        missing, unexpected = layer.load_state_dict({k[len("model.layers.0."):]: v for k, v in W.items()}, strict=False)
        assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)
        layer = layer.eval().to(dt)          # rounds the parameters and the cached rope tables (demo.py:234)
        keep = (lambda t: t.numpy()) if dt == torch.float32 else _u16
        with torch.no_grad():
            mask_now, past = am.clone(), None
            for s, x in enumerate(xs):
                x = x.to(dt)
                T = x.shape[1]
                pl = 0 if past is None else past[0].shape[2]
                # LlamaModel._prepare_decoder_attention_mask uses no state of the model (:475-496)
                m4 = ref.LlamaModel._prepare_decoder_attention_mask(None, mask_now, (2, T), x, pl)
                pos = mask_now.long().cumsum(-1) - 1               # prepare_inputs_for_generation :805-810
                pos.masked_fill_(mask_now == 0, 1)
                pos = pos[:, -T:]
                o = layer(x, attention_mask=m4, position_ids=pos, past_key_value=past, use_cache=True)
                out[f"h{s}_{tag}"] = keep(o[0])
                past = o[1]
                mask_now = torch.cat([mask_now, mask_now.new_ones(2, 1)], -1)
            out[f"k_{tag}"] = keep(past[0][:, list(REAL_LAYER_HEADS)])
            out[f"v_{tag}"] = keep(past[1][:, list(REAL_LAYER_HEADS)])
        del layer
    np.savez_compressed(os.path.join(OUT, "llama_real_layer.npz"), **out)
    print("llama_real_layer:", {k: v.shape for k, v in out.items()})


def make_qformer_real():
    """the reference's BertLMHeadModel(...).bert at its REAL shape (768 x 12 layers x 12 heads, 32 queries, cross-attention
    to 196 x 1408 every 2nd layer; Qformer.py:804-965), B = 1. Output only; input and weights regenerated by the test."""
    ref = _qformer_ref()
    q = QFormerCfg()
    bc = ref.BertConfig(vocab_size=30523, hidden_size=q.hidden, num_hidden_layers=q.layers, num_attention_heads=q.heads,
                        intermediate_size=q.inter, layer_norm_eps=q.ln_eps, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    bc.encoder_width = q.enc_width
    bc.add_cross_attention = True
    bc.cross_attention_freq = q.cross_freq
    bc.query_length = q.n_query
    model = ref.BertLMHeadModel(bc)
    W = synth.make_weights(synth.qformer_specs(q))
    sd = {k[len("Qformer."):]: v for k, v in W.items() if k.startswith("Qformer.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    for mk in missing:
        assert any(s in mk for s in ("word_embeddings", "position_embeddings", "position_ids", ".intermediate.dense", ".output.dense", ".output.LayerNorm", "cls.")), mk
    model.eval()
    img = synth.synth("golden.qf_real_img", (1, 196, q.enc_width), -2.0, 2.0)
    with torch.no_grad():
        o = model.bert(query_embeds=W["query_tokens"].expand(1, -1, -1), encoder_hidden_states=img,
                       encoder_attention_mask=torch.ones(1, 196, dtype=torch.long), return_dict=True)
    np.savez_compressed(os.path.join(OUT, "qformer_real.npz"), out=o.last_hidden_state.numpy())
    print("qformer_real:", o.last_hidden_state.shape, float(o.last_hidden_state.abs().mean()))


def make_projector():
    ref = _load("biovil_t/modules.py", "ref_modules")
    v = VisionCfg(img=128, stem=32, planes=(32, 64, 128, 256), blocks=(1, 2, 2, 1), b2v=64, proj=352)
    mlp = ref.MLP(input_dim=2 * v.b2v, output_dim=v.proj, hidden_dim=v.proj, use_1x1_convs=True)
    W = synth.make_weights(synth.vision_specs(v))
    J = "visual_encoder.projector."
    sd = {k[len(J):]: t for k, t in W.items() if k.startswith(J)}
    sd["model.1.num_batches_tracked"] = torch.tensor(0)
    mlp.load_state_dict(sd)
    mlp.eval()
    x = synth.synth("golden.patch_fused", (2, 2 * v.b2v, v.grid, v.grid), -1.5, 1.5)
    with torch.no_grad():
        pp = mlp(x)
        # blip2_qformer.py:469 + blip2.py:199-205, with torch's own LayerNorm as the reference uses it
        ln = torch.nn.LayerNorm(v.proj)
        ln.weight.data.copy_(W["ln_vision.weight"])
        ln.bias.data.copy_(W["ln_vision.bias"])
        emb = ln(pp.reshape(2, -1, v.proj).float())
    np.savez_compressed(os.path.join(OUT, "projector.npz"), x=x.numpy(), projected=pp.numpy(), image_embeds=emb.numpy())
    print("projector:", pp.shape, emb.shape)


def make_vit_pooler():
    import types
    import torch.nn as nn

    class Mlp(nn.Module):                     # timm==0.4.12 timm/models/layers/mlp.py, restated (third-party, absent)
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class DropPath(nn.Module):                # stochastic depth: the identity in eval mode
        def __init__(self, drop_prob=None):
            super().__init__()

        def forward(self, x):
            return x

    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    timm.models.layers = types.ModuleType("timm.models.layers")
    timm.models.layers.Mlp, timm.models.layers.DropPath = Mlp, DropPath
    timm.models.layers.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules.update({"timm": timm, "timm.models": timm.models, "timm.models.layers": timm.models.layers})
    try:
        ref = _load("biovil_t/transformer.py", "ref_transformer")
    finally:
        for k in ("timm", "timm.models", "timm.models.layers"):
            sys.modules.pop(k, None)
    v = VisionCfg(img=128, stem=32, planes=(32, 64, 128, 256), blocks=(1, 2, 2, 1), b2v=64, proj=352, pool_blocks=2, pool_heads=2)
    g, C = v.grid, v.b2v
    pooler = ref.VisionTransformerPooler(input_dim=C, grid_shape=(g, g), num_heads=v.pool_heads, num_blocks=v.pool_blocks)
    W = synth.make_weights(synth.vision_specs(v))
    P = "visual_encoder.encoder.vit_pooler."
    sd = {k[len(P):]: t for k, t in W.items() if k.startswith(P)}
    missing, unexpected = pooler.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)          # pos_embed is a non-persistent buffer
    pooler.eval()
    cur = synth.synth("golden.pool_cur", (2, C, g, g), -1.5, 1.5)
    prev = synth.synth("golden.pool_prev", (2, C, g, g), -1.5, 1.5)
    with torch.no_grad():
        out = pooler(current_image=cur, previous_image=prev)
    np.savez_compressed(os.path.join(OUT, "vit_pooler.npz"), cur=cur.numpy(), prev=prev.numpy(), out=out.numpy(),
                        pos_embed=pooler.pos_embed.numpy())
    print("vit_pooler:", out.shape, float(out.abs().mean()))


def make_downstream():
    """downstream_tasks/*.py prompt builders (the correction module imports the CheXbert labeller at module level -- stubbed)."""
    import json
    import types
    stub = types.ModuleType("chexbert.run_chexbert")
    stub.run_chexbert_labeler = lambda *a, **k: None
    pkg = types.ModuleType("chexbert")
    sys.modules.update({"chexbert": pkg, "chexbert.run_chexbert": stub})
    try:
        corr = _load("downstream_tasks/automated_correction.py", "ref_corr")
        cls = _load("downstream_tasks/chexpert_classification_downstream.py", "ref_cls")
    finally:
        sys.modules.pop("chexbert", None)
        sys.modules.pop("chexbert.run_chexbert", None)
    cols = ["No Finding", "Enlarged Cardiomediastinum", "Cardiomegaly", "Lung Opacity", "Lung Lesion", "Edema", "Consolidation",
            "Pneumonia", "Atelectasis", "Pneumothorax", "Pleural Effusion", "Pleural Other", "Fracture", "Support Devices"]
    hist = ["A chat. USER: Image information: X. write the report ASSISTANT:The heart is enlarged. No effusion.",
            "A chat. USER: q ASSISTANT:Clear lungs.", "A chat. USER: q ASSISTANT:Right pneumothorax with chest tube.",
            "A chat. USER: q ASSISTANT:Stable."]
    preds = np.zeros((4, 14), dtype=np.int64)
    labels = np.zeros((4, 14), dtype=np.int64)
    preds[0, [2, 10]] = 1; labels[0, [2, 5, 8]] = 1            # fp: effusion; fn: edema, atelectasis
    preds[1, [0]] = 1; labels[1, [3, 6, 7]] = 1                 # fp: only No Finding (ignored); fn: three
    preds[2, [9, 13, 12]] = 1; labels[2, [9]] = 1               # fp: two; no fn
    preds[3, [0]] = 1; labels[3, [0]] = 1                       # nothing to correct
    gold = {"cols": cols, "history": hist, "preds": preds.tolist(), "labels": labels.tolist(),
            "correction_prompts": corr.get_correction_prompts(list(hist), cols, preds, labels),
            "correction_labels": [list(map(list, x)) for x in corr.get_correction_labels(cols, preds, labels)],
            "bin_prompts": cls.get_chexpert_prompts_bin(list(hist), cols), "all_prompts": cls.get_chexpert_prompts_all(list(hist), cols)}
    with open(os.path.join(OUT, "downstream.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("downstream:", len(gold["correction_prompts"]), len(gold["bin_prompts"][0]))


def make_prompter():
    """utils/prompter.py Prompter('vicuna_v11') outputs (the reference resolves data/templates relative to the CWD)."""
    import json
    ref = _load("utils/prompter.py", "ref_prompter")
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        pr = ref.Prompter("vicuna_v11")
    finally:
        os.chdir(cwd)
    cases = [("Write a report.", None, None), ("Instruction", "some input", None), ("Q?", None, " label text"),
             ("A " + "<IMG>" * 32 + " B", "", None)]
    gold = {"generate_prompt": [[list(c), pr.generate_prompt(*c)] for c in cases],
            "get_response": [[s, pr.get_response(s)] for s in
                             ["USER: hi ASSISTANT: hello there ", "x ASSISTANT: a ASSISTANT:  b  ", "no split here"]]}
    with open(os.path.join(OUT, "prompter.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("prompter:", len(gold["generate_prompt"]), len(gold["get_response"]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["llama", "qformer", "projector", "prompter", "vit_pooler", "downstream", "llama_real_layer", "qformer_real"]
    if "llama" in which:
        make_llama()
    if "qformer" in which:
        make_qformer()
    if "projector" in which:
        make_projector()
    if "prompter" in which:
        make_prompter()
    if "vit_pooler" in which:
        make_vit_pooler()
    if "downstream" in which:
        make_downstream()
    if "llama_real_layer" in which:
        make_llama_real_layer()
    if "qformer_real" in which:
        make_qformer_real()

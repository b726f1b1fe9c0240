"""CPU check of the load-time weight preparation (radialog_amd/weights.py): the engine tensors (BN folded, K re-ordered
to (kh,kw,c), stem padded to 7x8x4, missing_previous_emb folded into a bias, fused QKV / cross-KV, interleaved gate/up)
are pushed through a plain fp32 torch emulation of the launch sequence librdx runs (api.hip) and must reproduce the
oracle. This pins the data layout contract between Python and the HIP library without needing a GPU."""
import torch
import torch.nn.functional as F

from oracle import ref_cpu
from radialog_amd import synth, weights as W
from radialog_amd.config import small_cfg
from radialog_amd._lib import RDX_W_GEMM


def _items(it):
    return {name: t.float() for name, t, _ in it}


def _conv_rows(x, KH, KW, stride, pad, Ho, Wo):
    """NHWC x -> im2col rows [B*Ho*Wo, KH*KW*C] with K ordered (kh, kw, c)."""
    xp = F.pad(x, (0, 0, pad, pad + KW, pad, pad + KH))
    cols = [xp[:, kh: kh + stride * Ho: stride, kw: kw + stride * Wo: stride, :] for kh in range(KH) for kw in range(KW)]
    return torch.cat(cols, dim=-1).reshape(-1, KH * KW * x.shape[-1])


def test_vision_and_qformer_engine_tensors_reproduce_oracle():
    cfg = small_cfg()
    v, q = cfg.vision, cfg.qformer
    Wt = synth.make_weights({**synth.vision_specs(v), **synth.qformer_specs(q)})
    get = Wt.__getitem__
    E = _items(W.vision_items(get, v))
    assert torch.equal(W.sine_pos_embed(v.grid, v.b2v), ref_cpu.sine_pos_embed(v.grid, v.b2v))
    E.update(_items(W.qformer_items(get, q)))
    B = 2
    img = synth.synth_images(B, v.img)
    ref_q, ref_emb = ref_cpu.forward_image(img, Wt, cfg)

    # stem: NHWC4 image padded by 3, 7x8 taps (the 8th is zero), stride 2
    S = v.img
    x = F.pad(img.permute(0, 2, 3, 1), (0, 1))                              # [B,S,S,4]
    Hc = S // 2
    rows = _conv_rows(x, 7, 8, 2, 3, Hc, Hc)
    x = F.relu(rows @ E["v.conv1.w"].t() + E["v.conv1.b"]).reshape(B, Hc, Hc, v.stem)
    x = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    Hc, C = S // 4, v.stem
    for li, nblk in enumerate(v.blocks, start=1):
        for blk in range(nblk):
            p = f"v.l{li}.{blk}."
            planes = v.planes[li - 1]
            stride = 2 if (blk == 0 and li > 1) else 1
            Ho = Hc // stride
            t1 = F.relu(x.reshape(-1, C) @ E[p + "c1.w"].t() + E[p + "c1.b"]).reshape(B, Hc, Hc, planes)
            t2 = F.relu(_conv_rows(t1, 3, 3, stride, 1, Ho, Ho) @ E[p + "c2.w"].t() + E[p + "c2.b"])
            idt = x.reshape(-1, C)
            if blk == 0:
                idt = _conv_rows(x, 1, 1, stride, 0, Ho, Ho) @ E[p + "ds.w"].t() + E[p + "ds.b"]
            x = F.relu(t2 @ E[p + "c3.w"].t() + E[p + "c3.b"] + idt).reshape(B, Ho, Ho, 4 * planes)
            Hc, C = Ho, 4 * planes
    P = Hc * Hc
    t = x.reshape(-1, C) @ E["v.b2v.w"].t()
    t = F.relu(t @ E["v.proj1.w"].t() + E["v.proj1.b"])
    pp = (t @ E["v.proj2.w"].t() + E["v.proj2.b"]).reshape(B, P, v.proj)    # NHWC
    # scramble: token (r, c) = flat element r*C + c of the [C][P] matrix
    f = torch.arange(P * v.proj)
    tok = pp[:, f % P, f // P].reshape(B, P, v.proj)
    emb = F.layer_norm(tok, (v.proj,), E["v.ln.g"][0], E["v.ln.b"][0], v.ln_eps)
    assert torch.allclose(emb, ref_emb, rtol=0, atol=2e-3), float((emb - ref_emb).abs().max())

    # Q-Former with the fused tensors
    H, NQ, nh = q.hidden, q.n_query, q.heads
    xq = E["q.query_ln"].expand(B, -1, -1).reshape(B * NQ, H)
    kvx = emb.reshape(B * P, -1) @ E["q.cross.wkv"].t() + E["q.cross.bkv"]

    def attn(qm, km, vm, Tk):
        d = H // nh
        qh = qm.reshape(B, NQ, nh, d).transpose(1, 2)
        kh = km.reshape(B, Tk, nh, d).transpose(1, 2)
        vh = vm.reshape(B, Tk, nh, d).transpose(1, 2)
        p = torch.softmax(qh @ kh.transpose(-1, -2) / d ** 0.5, -1)
        return (p @ vh).transpose(1, 2).reshape(B * NQ, H)

    ci = 0
    for l in range(q.layers):
        o = f"q{l}."
        qkv = xq @ E[o + "self.wqkv"].t() + E[o + "self.bqkv"]
        ctx = attn(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], NQ)
        xq = F.layer_norm(ctx @ E[o + "self.wo"].t() + E[o + "self.bo"] + xq, (H,), E[o + "self.ln_g"][0], E[o + "self.ln_b"][0], q.ln_eps)
        if q.has_cross(l):
            qq = xq @ E[o + "cross.wq"].t() + E[o + "cross.bq"]
            kk, vv = kvx[:, ci * 2 * H: ci * 2 * H + H], kvx[:, ci * 2 * H + H: (ci + 1) * 2 * H]
            ci += 1
            ctx = attn(qq, kk, vv, P)
            xq = F.layer_norm(ctx @ E[o + "cross.wo"].t() + E[o + "cross.bo"] + xq, (H,), E[o + "cross.ln_g"][0], E[o + "cross.ln_b"][0], q.ln_eps)
        h = F.gelu(xq @ E[o + "ffn.w1"].t() + E[o + "ffn.b1"])
        xq = F.layer_norm(h @ E[o + "ffn.w2"].t() + E[o + "ffn.b2"] + xq, (H,), E[o + "ffn.ln_g"][0], E[o + "ffn.ln_b"][0], q.ln_eps)
    out = xq.reshape(B, NQ, H)
    assert torch.allclose(out, ref_q, rtol=0, atol=2e-3), float((out - ref_q).abs().max())


def test_llama_fused_layouts():
    cfg = small_cfg().llama
    Wt = synth.make_weights(synth.llama_specs(cfg, lora=True))
    items = {n: (t, k) for n, t, k in W.llama_items(Wt.__getitem__, cfg, lora=True)}
    H, I, r = cfg.hidden, cfg.inter, cfg.lora_r
    wqkv, kind = items["l0.wqkv"]
    assert kind == RDX_W_GEMM and wqkv.shape == (3 * H + 2 * r, H)
    assert torch.equal(wqkv[H:2 * H], Wt["model.layers.0.self_attn.k_proj.weight"])
    assert torch.equal(wqkv[3 * H + r:], Wt["model.layers.0.self_attn.v_proj.lora_A.weight"])
    wgu = items["l1.wgu"][0]
    assert wgu.shape == (2 * I, H)
    t = 5                                           # tile t: rows 16t..16t+7 gate[8t..], rows 16t+8.. up[8t..]
    assert torch.equal(wgu[16 * t: 16 * t + 8], Wt["model.layers.1.mlp.gate_proj.weight"][8 * t: 8 * t + 8])
    assert torch.equal(wgu[16 * t + 8: 16 * t + 16], Wt["model.layers.1.mlp.up_proj.weight"][8 * t: 8 * t + 8])
    cos, _ = items["rope.cos"]
    ref_cos, _ = ref_cpu.rope_tables(cfg.head_dim, cfg.max_pos, cfg.rope_base, torch.float32)
    assert torch.equal(cos, ref_cos)


def test_classifier_items_cover_the_reference_state_dict():
    """ChexpertClassifier weights (`biovil_encoder.*`, fc1, fc2) -> engine tensors: every reference tensor is consumed, the
    head keeps its shapes, the odd-sized grid is computed like torch's conv arithmetic."""
    from radialog_amd import synth, weights
    from radialog_amd.config import classifier_cfg, small_classifier_cfg
    assert classifier_cfg().vision.grid == 16 and small_classifier_cfg().vision.grid == 5
    cfg = small_classifier_cfg()
    W = synth.make_weights(synth.classifier_specs(cfg.vision, cfg.cls))
    used = set()

    def get(name):
        used.add(name)
        return W[name]

    items = {n: (t, k) for n, t, k in weights.classifier_items(get, cfg.vision, cfg.cls)}
    gp = cfg.vision.grid // cfg.cls.pool
    assert items["cls.fc1.w"][0].shape == (cfg.cls.hidden, cfg.vision.proj * gp * gp)
    assert items["cls.fc2.w"][0].shape == (cfg.cls.classes, cfg.cls.hidden)
    assert "v.ln.g" not in items and "v.proj2.w" in items
    unused = {k for k in W if k not in used and "vit_pooler" not in k and "num_batches" not in k}
    assert not unused, sorted(unused)[:5]

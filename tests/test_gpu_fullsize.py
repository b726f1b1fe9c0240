"""Parity at the sizes BASELINE.json's configs are quoted on -- the shapes bench.py actually runs:

  * the FULL-SIZE image encoder (448 px, ResNet-50 3-4-6-3 with a 64-wide stem, 2048 trunk channels, projector 1408, Q-Former
    768 x 12) at batch 1 (split-K over the deep layers' few tiles) and batch 8 (256^2-tile / batched dispatch), against
    oracle/ref_cpu.forward_image (blip2_qformer.py:467-484);
  * the FULL-DEPTH decoder: all 32 Vicuna-7B layers at production width, fp16 (the reference's dtype), batch 1, T = 160, greedy
    tokens and per-step logits against the oracle AND against the exactly-accumulated evaluation of the same rounding points;
  * image -> tokens END TO END: oracle image -> oracle fp32 Q-Former output -> oracle tokens versus HIP image -> HIP Q-Former
    output -> HIP tokens (no hand-over of the GPU's embeddings to the oracle's decoder).
"""
import pytest
import torch

from radialog_amd import synth
from radialog_amd.config import LlamaCfg, RaDialogCfg, full_cfg, small_cfg
from _parity import check_greedy, teacher_forced

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16}
ENC_TOL = {"f16": 5e-3, "bf16": 3e-2}


def _rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def full_vis_w():
    cfg = full_cfg()
    return cfg, synth.make_weights({**synth.vision_specs(cfg.vision), **synth.qformer_specs(cfg.qformer)})


@pytest.fixture(scope="module")
def full_enc_ref(full_vis_w):
    from oracle import ref_cpu
    cfg, W = full_vis_w
    img = synth.synth_images(8, cfg.vision.img)
    with torch.no_grad():
        ref_q, ref_emb = ref_cpu.forward_image(img, W, cfg)
    return img, ref_q, ref_emb


@pytest.mark.parametrize("dtype,trunk", [("f16", "packed"), ("bf16", "packed"), ("f16", "rowmajor")])
def test_full_size_encoder_matches_oracle(full_vis_w, full_enc_ref, dtype, trunk, monkeypatch):
    """trunk = "packed": the round-4 default -- the ResNet trunk on fragment-packed activations (pconv_k); "rowmajor" (RDX_PCONV=0 at
    rdx_create): the LDS-staged row-major kernels of rounds 1-3, kept as the A/B leg. Both against the fp32 oracle, and against each other
    within the same tolerance."""
    from radialog_amd.engine import RdxEngine, synth_getter
    cfg, W = full_vis_w
    img, ref_q, ref_emb = full_enc_ref
    if trunk == "rowmajor":
        monkeypatch.setenv("RDX_PCONV", "0")
    eng = RdxEngine(cfg, dtype=dtype, device=0, llama=False)
    eng.load_weights(synth_getter(cfg, eng.device), llama=False)
    tol = ENC_TOL[dtype]
    for B in (1, 8, 3):                   # 1: few-tile grids; 8: batched tiles; 3 after 8: a smaller batch in the grown workspace
        q, emb = eng.encode_image(img[:B].to(eng.device))
        assert q.shape == (B, 32, 768) and emb.shape == (B, 196, 1408)
        assert torch.isfinite(q).all() and torch.isfinite(emb).all()
        e_emb, e_q = _rel_l2(emb.cpu(), ref_emb[:B]), _rel_l2(q.cpu(), ref_q[:B])
        print(f"full-size encoder {dtype} {trunk} B={B}: rel-L2 image_embeds {e_emb:.3e}, Q-Former out {e_q:.3e}")
        assert e_emb < tol, f"B={B}: image_embeds (ResNet-50 trunk + projector + scramble + ln_vision) rel-L2 {e_emb}"
        assert e_q < tol, f"B={B}: Q-Former last_hidden_state rel-L2 {e_q}"
        # per-image: no row may hide behind the batch norm
        for b in range(B):
            assert _rel_l2(q[b].cpu(), ref_q[b]) < 2 * tol, f"B={B} image {b}"
    eng.close()


def test_image_to_tokens_end_to_end_matches_oracle():
    """Nothing crosses over: the oracle runs image -> fp32 encoder -> fp32 Q-Former -> fp16 decoder -> tokens on the CPU, the engine
    runs the same chain on the GPU (fp16 encoder with fp32 accumulation). The encoder's rounding noise (rel-L2 ~1e-3) reaches the
    decoder through img_proj and the 32 spliced rows; tokens must still be the oracle's and logits within the decoder tolerance
    plus what that input perturbation is worth (measured and asserted: < 2e-2)."""
    from oracle import ref_cpu
    from radialog_amd.engine import RdxEngine, synth_getter
    cfg = small_cfg()
    W = synth.make_weights({**synth.vision_specs(cfg.vision), **synth.qformer_specs(cfg.qformer), **synth.llama_specs(cfg.llama)})
    B, T, N = 3, 64, 12
    img = synth.synth_images(B, cfg.vision.img, seed=31)
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=5, pad_rows=False, seed=17)
    ids[1] = torch.cat([torch.zeros(4, dtype=torch.long), ids[1, : T - 4]])
    with torch.no_grad():
        rq, _ = ref_cpu.forward_image(img, W, cfg)
        ref = ref_cpu.LlamaOracle(W, cfg.llama, torch.float16, lora=True).generate_greedy(ids, rq, max_new=N, eos_id=-1)
    eng = RdxEngine(cfg, dtype="f16", device=0, max_batch=B, max_len=128)
    eng.load_weights(synth_getter(cfg, eng.device))
    q, _ = eng.encode_image(img.to(eng.device), want_image_embeds=False)
    toks, scores, n = eng.generate(ids, q, max_new=N, eos_id=-1, output_scores=True)
    cmp_, tot, worst = check_greedy(toks, scores, ref, 2e-2, 0.9, "image -> tokens")
    print(f"end to end: {cmp_}/{tot} pairs identical, worst logit error {worst:.4g}, encoder rel-L2 {_rel_l2(q.cpu(), rq):.3e}")
    eng.close()


def test_full_size_encoder_into_production_width_decoder_end_to_end(full_vis_w):
    """The same end-to-end identity at the benchmark's shapes: full-size encoder (768-wide Q-Former output) into a production-width
    decoder (hidden 4096, inter 11008, vocab 32001; two layers so that the CPU oracle finishes in seconds), fp16."""
    from oracle import ref_cpu
    from radialog_amd.engine import RdxEngine, synth_getter
    base, Wv = full_vis_w
    cfg = RaDialogCfg(llama=LlamaCfg(layers=2), qformer=base.qformer, vision=base.vision)
    Wl = synth.make_weights(synth.llama_specs(cfg.llama, lora=True))
    B, T, N = 2, 96, 6
    img = synth.synth_images(B, cfg.vision.img, seed=77)
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=20, pad_rows=False, seed=19)
    with torch.no_grad():
        rq, _ = ref_cpu.forward_image(img, Wv, cfg)
        ref = ref_cpu.LlamaOracle(Wl, cfg.llama, torch.float16, lora=True).generate_greedy(ids, rq, max_new=N, eos_id=-1)
    eng = RdxEngine(cfg, dtype="f16", device=0, max_batch=B, max_len=128)
    eng.load_weights(synth_getter(cfg, eng.device))
    q, _ = eng.encode_image(img.to(eng.device), want_image_embeds=False)
    toks, scores, n = eng.generate(ids, q, max_new=N, eos_id=-1, output_scores=True)
    # the full-size encoder's fp16 rounding noise (Q-Former output rel-L2 ~2e-3, twice the small model's) enters through 32 rows
    cmp_, tot, worst = check_greedy(toks, scores, ref, 3e-2, 0.9, "full-size image -> tokens")
    print(f"full-size end to end: {cmp_}/{tot} pairs identical, worst logit error {worst:.4g}, encoder rel-L2 {_rel_l2(q.cpu(), rq):.3e}")
    eng.close()


def _full_depth_engine_and_weights(dtype, B, max_len):
    """All 32 production-width layers: the engine, and the SAME bytes for the oracle -- generated on the GPU (seconds instead of
    minutes), rounded to the model dtype, moved to the host."""
    from radialog_amd.engine import RdxEngine, synth_getter
    cfg = full_cfg()
    dt = DT[dtype]
    eng = RdxEngine(cfg, dtype=dtype, device=0, max_batch=B, max_len=max_len, vision=False)
    eng.load_weights(synth_getter(cfg, eng.device), vision=False)
    specs = synth.llama_specs(cfg.llama, lora=True)
    probe = "model.layers.0.self_attn.q_proj.lora_A.weight"
    assert torch.equal(specs[probe][1](probe, specs[probe][0], "cpu"), specs[probe][1](probe, specs[probe][0], eng.device).cpu())
    W = {name: gen(name, shape, eng.device).to(dt).cpu() for name, (shape, gen) in specs.items()}
    return cfg, eng, W


def _head(ref, n):
    """The first n steps of a LlamaOracle.generate_greedy result (the free-running legs compare a prefix of the long oracle run)."""
    return {"tokens": ref["tokens"][:, :n], "scores": ref["scores"][:n], "margins": ref["margins"][:n]}


FULL_TOL = {"f16": 6e-2, "bf16": 0.45}      # one-layer tolerance x sqrt(32 layers): 1e-2 -> 6e-2 (fp16, north_star's dtype), 8e-2 -> 0.45 (bf16)


@pytest.mark.slow
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_full_depth_32_layer_decoder_matches_oracle(dtype):
    """BASELINE configs[0]/[1] decoder in full: 32 layers at production width, batch 1, the bench's 160-token prompt with the 32 <IMG>
    slots -- in fp16 (the reference's dtype) AND in bf16 (the dtype bench.py times). Two legs on one engine:

    (a) 6 free-running greedy tokens through the hipGraph step. Three evaluations of the same op sequence and rounding points: HIP
        (fp32 MFMA accumulation), the torch-CPU oracle (fp32 accumulation in torch's order) and the exact one (fp64 accumulation).
        Over 32 layers the accumulation-order noise of ANY two implementations exceeds the one-layer tolerance (the oracle itself is
        that far from the exact evaluation), so the bar is: tokens identical to the oracle's wherever its margin exceeds twice the
        measured logit error (tests/_parity.py), and the HIP logits no further from the exact evaluation than 1.5 x the oracle's own
        distance (+ 1 ulp at |logit| in [4, 8)); the distances are printed.
    (b) round 4 (VERDICT r3 "weak" 1): a REPORT-LENGTH horizon -- 64 teacher-forced decode steps (positions 160 .. 223; the oracle's
        token is fed back through rdx_decode_step_ids so a near-tie flip cannot end the comparison) at full depth, with an ABSOLUTE
        logit bar: 99 % of the steps within 1e-2 x sqrt(32) = 6e-2 in fp16 (0.45 in bf16), all within 1.5 x that, every differing argmax
        at an oracle margin <= 2 x the measured error of that step, and >= 90 % (fp16) / 75 % (bf16) argmax identity."""
    from oracle import ref_cpu
    dt = DT[dtype]
    cfg, eng, W = _full_depth_engine_and_weights(dtype, 1, 256)
    T, N, NTF = 160, 6, 64
    ids = synth.synth_prompt_ids(1, T, vocab=cfg.llama.vocab)
    qf = synth.synth("t.qf_full", (1, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    toks, scores, n = eng.generate(ids, qf, max_new=N, eos_id=-1, output_scores=True)
    toks, scores = toks.cpu().long().clone(), scores.float().cpu().clone()
    with torch.no_grad():
        ref = ref_cpu.LlamaOracle(W, cfg.llama, dt, lora=True).generate_greedy(ids, qf, max_new=NTF, eos_id=-1)
        truth = ref_cpu.LlamaOracle(W, cfg.llama, dt, lora=True, exact=True).generate_greedy(ids, qf, max_new=3, eos_id=-1)
    del W
    def dist(a, a_tok, b, b_tok, steps):
        w = 0.0
        for s in range(steps):
            w = max(w, float((a[s][0].float() - b[s][0].float()).abs().max()))
            if int(a_tok[0, s]) != int(b_tok[0, s]):
                break
        return w
    e_ho = dist(scores, toks, ref["scores"], ref["tokens"], N)
    e_ht = dist(scores, toks, truth["scores"], truth["tokens"], 3)
    e_ot = dist(ref["scores"], ref["tokens"], truth["scores"], truth["tokens"], 3)
    print(f"full depth {dtype}: tokens hip {toks[0].tolist()} oracle {ref['tokens'][0, :N].tolist()} exact {truth['tokens'][0].tolist()}; "
          f"margins {[round(float(m), 3) for m in ref['margins'][:N, 0]]}; |hip-oracle| {e_ho:.4g} |hip-exact| {e_ht:.4g} |oracle-exact| {e_ot:.4g}")
    # six pairs cannot carry a percentage bar (one near-tie ends the row): the exact-evaluation bound below is the quantitative bar of leg (a)
    check_greedy(toks, scores, _head(ref, N), FULL_TOL[dtype], 0.0, f"full depth {dtype}")
    ulp = 2.0 ** -8 if dtype == "f16" else 2.0 ** -5
    assert e_ht <= 1.5 * e_ot + ulp, f"HIP is {e_ht:.4g} from the exact evaluation, the torch-CPU oracle only {e_ot:.4g}"
    # leg (b): 64 teacher-forced steps
    same, total, worst = teacher_forced(eng, ref, ids, qf, NTF, FULL_TOL[dtype], f"full depth teacher-forced {dtype}")
    eng.close()
    print(f"full depth {dtype}, {NTF} teacher-forced steps (positions {T}..{T + NTF - 1}): {same}/{total} argmax tokens identical, worst logit error "
          f"{worst:.4g} (bar: 99 % < {FULL_TOL[dtype]}, all < {1.5 * FULL_TOL[dtype]:.3g}); median oracle margin {float(ref['margins'].median()):.3g}")
    assert same >= {"f16": 0.9, "bf16": 0.75}[dtype] * total, f"{dtype}: only {same}/{total} full-depth steps chose the oracle's token"


@pytest.mark.slow
def test_full_depth_batch32_decoder_fp16_matches_oracle():
    """BASELINE configs[2] decoder in full: 32 layers at production width, batch 32 with left-padded rows (prompts as the bench builds them,
    every 4th row padded; T = 96 instead of the bench's 160 since round 4 (the padded rows' 32-slot image block needs T >= 92): the oracle's batched prefill is most of this test's minutes and
    the whole -m gpu suite has to fit the driver's 20-minute step on a slower host -- positions 96 .. 103 run the same kernels; bench.py's `parity_b32` checks row 0 of the timed T = 160 run at full depth) -- the
    activation-stationary / K-split kernels, the throughput attention with the row-major K cache and the batched prefill GEMMs at full
    depth. fp16, the reference's dtype. Two legs on one engine and ONE oracle run (8 greedy tokens):
    (a) 4 free-running tokens through the hipGraph-captured step against the oracle's first 4: logits within 6e-2 (1e-2 x sqrt(32
        layers)), tokens identical wherever the oracle's margin exceeds twice the measured error, >= 90 % of the 64 pairs (16 oracle rows);
    (b) round 4: teacher-forced steps (round 5: 8 of them = 128 (row, step) pairs, none lost to a near-tie -- the suite grew by the 3-16 / 33-128 row
        legs and bench.py's `parity_b32` now checks row 0 of the timed T = 160 batch-32 run at full depth over 32 steps), same absolute bar, >= 90 % argmax identity."""
    from oracle import ref_cpu
    cfg, eng, W = _full_depth_engine_and_weights("f16", 32, 128)
    B, T, N, NTF = 32, 96, 4, 8
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=True, seed=7)
    qf = synth.synth("t.qf_full32", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    toks, scores, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0, output_scores=True, use_graph=True)
    # round 5 (the suite has to stay well inside the driver's 20 minutes): the engine runs all 32 rows, the ORACLE half of them -- rows 0-3, 8-11, 16-19, 24-27: both
    # row tiles, every left-pad phase; a row's arithmetic does not depend on its neighbours. The other rows ride along and are checked for NaN.
    rows = [r for r in range(B) if (r >> 2) % 2 == 0]
    toks, scores = toks.cpu().long()[rows].clone(), scores.float().cpu()[:, rows].clone()
    with torch.no_grad():
        ref = ref_cpu.LlamaOracle(W, cfg.llama, torch.float16, lora=True).generate_greedy(ids[rows], qf[rows], max_new=NTF, eos_id=-1, pad_id=0)
    del W
    cmp_, tot, worst = check_greedy(toks, scores, _head(ref, N), 6e-2, 0.9, "full depth batch 32 fp16")
    print(f"full depth batch 32 fp16: {cmp_}/{tot} pairs identical, worst logit error {worst:.4g}, "
          f"smallest oracle margin {float(ref['margins'][:N].min()):.4g}")
    same, total, worst = teacher_forced(eng, ref, ids, qf, NTF, 6e-2, "full depth batch 32 teacher-forced fp16", rows=rows)
    eng.close()
    print(f"full depth batch 32 fp16, {NTF} teacher-forced steps: {same}/{total} argmax tokens identical, worst logit error {worst:.4g}")
    assert same >= 0.9 * total

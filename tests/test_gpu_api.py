"""The reference-shaped Python surface end to end on the GPU (small shapes): Blip2Qformer.forward_image ->
{dicom: emb} / current_chat_img.pt hand-off -> LlamaForCausalLM.generate -> .sequences / .scores, checked against the
oracle driven the same way."""
import os

import pytest
import torch

from radialog_amd import synth
from radialog_amd.config import small_cfg

pytestmark = pytest.mark.gpu


def test_forward_image_then_generate_like_demo_and_test_py(tmp_path, monkeypatch):
    from oracle import ref_cpu
    from radialog_amd.blip2_qformer import Blip2Qformer
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM
    monkeypatch.chdir(tmp_path)
    cfg = small_cfg()
    blip = Blip2Qformer(img_size=cfg.vision.img, dtype="f16", cfg=cfg, synthetic=True).to(torch.device("cuda")).eval()
    img = synth.synth_images(2, cfg.vision.img)
    q, emb = blip.forward_image(img.cuda())
    assert q.shape == (2, 32, cfg.qformer.hidden) and emb.shape == (2, cfg.vision.n_patches, cfg.vision.proj)
    assert q.dtype == torch.float32 and q.is_cuda

    lm = LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.float16, cfg=cfg.llama, max_batch=2, max_len=128, synthetic=True).eval()
    # test.py path: embeddings by dicom id
    lm.model.blip_embeddings.update({"a": q[0].cpu().numpy(), "b": q[1].cpu().numpy()})
    ids = synth.synth_prompt_ids(2, 48, vocab=cfg.llama.vocab, img_offset=4)
    ids[1] = torch.cat([torch.zeros(3, dtype=torch.long), ids[1, :45]])
    out = lm.generate(input_ids=ids, dicom=["a", "b"], return_dict_in_generate=True, output_scores=True, max_new_tokens=6,
                      eos_token_id=-1)
    assert out.sequences.shape == (2, 54) and out.sequences.dtype == torch.int64
    assert torch.equal(out.sequences[:, :48].cpu(), ids)
    assert len(out.scores) == 6 and out.scores[0].shape == (2, cfg.llama.vocab)
    # demo.py path: current_chat_img.pt + use_img
    torch.save(q[:1].cpu(), "current_chat_img.pt")
    out1 = lm.generate(input_ids=ids[:1], dicom=None, use_img=True, return_dict_in_generate=True, output_scores=True,
                       max_new_tokens=6, eos_token_id=-1)
    assert torch.equal(out1.sequences[0], out.sequences[0])
    # oracle, same weights and the same embeddings
    W = synth.make_weights(synth.llama_specs(cfg.llama))
    ref = ref_cpu.LlamaOracle(W, cfg.llama, torch.float16).generate_greedy(ids, q.cpu(), max_new=6, eos_id=-1)
    from _parity import check_greedy
    check_greedy(out.sequences[:, 48:], torch.stack(out.scores), ref, 1e-2, 0.9, "reference-shaped surface")
    # plain ids without return_dict
    seq = lm.generate(input_ids=ids, dicom=["a", "b"], max_new_tokens=2, eos_token_id=-1)
    assert torch.is_tensor(seq) and seq.shape == (2, 50)


def test_findings_classifier_matches_oracle():
    """§8f rank 3: ChexpertClassifier inference (trunk + stock projector at an odd-sized grid, avg_pool2d, fc1, fc2)
    through rdx_classify_findings against the fp32 CPU oracle; predicted label sets must agree wherever the oracle's
    logit is not within tolerance of the decision boundary."""
    import torch
    from oracle import ref_cpu
    from radialog_amd import synth
    from radialog_amd.chexpert_model import ChexpertClassifier
    from radialog_amd.config import small_classifier_cfg
    cfg = small_classifier_cfg()
    W = synth.make_weights(synth.classifier_specs(cfg.vision, cfg.cls))
    img = synth.synth_images(3, cfg.vision.img, seed=5)
    with torch.no_grad():
        ref = ref_cpu.findings_logits(img, W, cfg.vision, cfg.cls)
    for dtype, tol in (("f16", 5e-3), ("bf16", 3e-2)):
        m = ChexpertClassifier(num_classes=cfg.cls.classes, cfg=cfg, dtype=dtype)
        m.load_state_dict(W)
        out = m(img.cuda()).float().cpu()
        assert out.shape == ref.shape and torch.isfinite(out).all()
        scale = float(ref.abs().max().clamp_min(1.0))
        err = float((out - ref).abs().max())
        assert err < 4 * tol * scale, f"{dtype}: logits differ by {err}"
        far = ref.abs() > 4 * tol * scale
        assert torch.equal((out > 0)[far], (ref > 0)[far])
        labels = m.predict_findings(img.cuda())
        assert len(labels) == 3 and all(isinstance(s, str) for s in labels)
        m._engine.close()


def test_embedding_dump_round_trip_feeds_generate_by_dicom(tmp_path, monkeypatch):
    """SURVEY.md 8f rank 1 -- the offline embedding dump (pretraining/train.py:134-173): batches of images through
    forward_image, `{dicom: float32[32, 768]}` pickled under pretraining/embs/, and the decoder picking them up by dicom id when
    it is constructed (modeling_llama_imgemb.py:454-462) must give exactly what passing the same Q-Former output directly gives."""
    import pickle
    from radialog_amd.blip2_qformer import Blip2Qformer
    from radialog_amd.embed_dump import dump_embeddings
    from radialog_amd.modeling_llama_imgemb import EMB_TEST_PKL, LlamaForCausalLM
    monkeypatch.chdir(tmp_path)
    cfg = small_cfg()
    blip = Blip2Qformer(img_size=cfg.vision.img, dtype="f16", cfg=cfg, synthetic=True).to(torch.device("cuda")).eval()
    N = 5
    img = synth.synth_images(N, cfg.vision.img, seed=40)
    dicoms = [f"d{i:03d}" for i in range(N)]
    os.makedirs(os.path.dirname(EMB_TEST_PKL))
    emb = dump_embeddings(img, dicoms, batch_size=2, model=blip, out_path=EMB_TEST_PKL)          # batches of 2, 2, 1
    with pytest.raises(ValueError):
        dump_embeddings(img, dicoms[:-1], model=blip)
    with pytest.raises(ValueError):
        dump_embeddings(img, dicoms)                                                          # no model and no synthetic opt-in
    on_disk = pickle.load(open(EMB_TEST_PKL, "rb"))
    assert sorted(on_disk) == dicoms and on_disk["d003"].shape == (32, cfg.qformer.hidden) and on_disk["d003"].dtype.name == "float32"
    direct = blip.forward_image(img.cuda())[0]
    for i, d in enumerate(dicoms):                                # batch composition must not change an image's embedding
        e = torch.from_numpy(emb[d]).double()
        assert float((e - direct[i].cpu().double()).norm() / e.norm()) < 3e-3, d        # fp16 tile / split-K choices differ with the batch
    lm = LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.float16, cfg=cfg.llama, max_batch=3, max_len=128, synthetic=True).eval()
    assert sorted(lm.model.blip_embeddings) == dicoms             # read from the pkl at construction, like the reference
    ids = synth.synth_prompt_ids(3, 48, vocab=cfg.llama.vocab, img_offset=4)
    pick = ["d004", "d001", "d002"]
    a = lm.generate(input_ids=ids, dicom=pick, return_dict_in_generate=True, output_scores=True, max_new_tokens=6, eos_token_id=-1)
    q = torch.stack([torch.from_numpy(emb[d]) for d in pick])
    b = lm.generate(input_ids=ids, qformer_embs=q, return_dict_in_generate=True, output_scores=True, max_new_tokens=6, eos_token_id=-1)
    assert torch.equal(a.sequences, b.sequences)
    assert all(torch.equal(x, y) for x, y in zip(a.scores, b.scores))
    with pytest.raises(KeyError):
        lm.generate(input_ids=ids, dicom=["d004", "nope", "d002"], max_new_tokens=2)


def test_rccl_allgather_inside_the_c_abi_single_rank():
    """rdx_comm_unique_id / rdx_comm_init / rdx_allgather_tokens (include/rdx.h) with a one-rank communicator -- the only size a
    one-GPU box can form; the two-rank data path is covered by tests/test_shard_gloo.py and by bench.py --gpus N under torchrun."""
    from radialog_amd import shard
    from radialog_amd._lib import RdxError
    from radialog_amd.engine import RdxEngine
    cfg = small_cfg()
    eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=2, max_len=64, vision=False)
    toks = torch.arange(2 * 7, dtype=torch.int32, device=eng.device).view(2, 7)
    with pytest.raises(RdxError):
        eng.allgather_tokens(toks)                                # no communicator yet
    shard.init_comm(eng, 0, 1)
    assert eng.comm_world == 1
    out = shard.allgather_tokens(toks, 1, engine=eng)
    assert out.data_ptr() != toks.data_ptr() and torch.equal(out, toks)
    with pytest.raises(RdxError):
        eng.comm_init(eng.comm_unique_id(), 0, 1)                 # already initialised
    eng.close()


def test_peft_wrapper_chat_turns_take_the_append_path():
    """demo.py sets `reuse_prefix_kv` on what init_vicuna returned; with --lora_model that is the PeftModelForCausalLM wrapper. The flag
    must reach the wrapped model: the second turn keeps the cached prefix (rdx_generate_append) instead of re-prefilling the whole
    conversation."""
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM, PeftModelForCausalLM
    cfg = small_cfg()
    inner = LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.float16, cfg=cfg.llama, max_batch=1, max_len=256, synthetic=True)
    lm = PeftModelForCausalLM(inner).eval()
    lm.reuse_prefix_kv = True
    qf = synth.synth("t.peftwrap", (1, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    ids1 = synth.synth_prompt_ids(1, 48, vocab=cfg.llama.vocab, img_offset=4)
    o1 = lm.generate(input_ids=ids1, qformer_embs=qf, return_dict_in_generate=True, max_new_tokens=5, eos_token_id=-1)
    assert inner._engine.last_kept_prefix == 0
    follow = torch.randint(3, 31999, (1, 9), generator=torch.Generator().manual_seed(3))
    ids2 = torch.cat([o1.sequences.cpu(), follow], dim=1)
    o2 = lm.generate(input_ids=ids2, qformer_embs=qf, return_dict_in_generate=True, max_new_tokens=4, eos_token_id=-1)
    assert inner._engine.last_kept_prefix == o1.sequences.shape[1] - 1
    fresh = LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.float16, cfg=cfg.llama, max_batch=1, max_len=256, synthetic=True).eval()
    f2 = fresh.generate(input_ids=ids2, qformer_embs=qf, return_dict_in_generate=True, max_new_tokens=4, eos_token_id=-1)
    assert torch.equal(o2.sequences, f2.sequences)
    inner._engine.close(); fresh._engine.close()


def test_set_weight_typed_is_bit_identical_to_the_widened_fp32_upload():
    """rdx_set_weight_typed (include/rdx.h): a checkpoint stored in fp16 goes to the library as it is. Same tokens and the same logit
    BITS as uploading the fp32-widened tensors (which is what round 2 did after inflating the whole checkpoint on the host)."""
    from radialog_amd.engine import RdxEngine, synth_getter
    cfg = small_cfg()
    ids = synth.synth_prompt_ids(2, 48, vocab=cfg.llama.vocab, img_offset=4)
    qf = synth.synth("t.typed", (2, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    outs = []
    for stored in (torch.float16, torch.bfloat16):
        for as_typed in (True, False):
            eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=2, max_len=128, vision=False)
            base = synth_getter(cfg, eng.device)
            get = (lambda n: base(n).to(stored)) if as_typed else (lambda n: base(n).to(stored).float())
            eng.load_weights(get, vision=False)
            toks, sc, n = eng.generate(ids, qf, max_new=6, eos_id=-1, output_scores=True)
            outs.append((toks.cpu().clone(), sc.cpu().clone()))
            eng.close()
        assert torch.equal(outs[-2][0], outs[-1][0]) and torch.equal(outs[-2][1].view(torch.int16), outs[-1][1].view(torch.int16))


def test_public_surface_fp8_configuration_matches_the_engine_run_directly():
    """BASELINE configs[4] through the reference's call surface: `LlamaForCausalLM.from_pretrained(..., weights_fp8=True)` (this build's one
    extra kwarg; the adapter stays the un-merged model-dtype epilogue of demo.py:232-234) must drive the same fp8 engine as
    RdxEngine(weights_fp8=True): identical tokens and score bits; and it must differ from the bf16 engine's logits (the flag is not ignored)."""
    from radialog_amd.engine import RdxEngine, synth_getter
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM
    cfg = small_cfg()
    B, T, N = 2, 48, 6
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=4)
    qf = synth.synth("t.pubfp8", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    outs = {}
    for fp8 in (True, False):
        lm = LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.bfloat16, cfg=cfg.llama, max_batch=B, max_len=128, synthetic=True,
                                              weights_fp8=fp8).eval()
        o = lm.generate(input_ids=ids, qformer_embs=qf, return_dict_in_generate=True, output_scores=True, max_new_tokens=N, eos_token_id=-1)
        assert lm._engine.weights_fp8 is fp8
        outs[fp8] = (o.sequences.cpu().clone(), torch.stack([s.cpu() for s in o.scores]))
        lm._engine.close()
    eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=128, lora=True, vision=False, weights_fp8=True)
    eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
    toks, sc, n = eng.generate(ids, qf, max_new=N, eos_id=-1, output_scores=True)
    assert torch.equal(outs[True][0][:, T:], toks.cpu().long()[:, :N])
    assert torch.equal(outs[True][1].view(torch.int16), sc.cpu()[:N].view(torch.int16))
    assert not torch.equal(outs[True][1], outs[False][1])
    eng.close()


@pytest.mark.parametrize("h,w", [(1000, 601), (601, 1000), (512, 512), (615, 512), (300, 400), (3056, 2544), (520, 700)])
def test_gpu_image_transform_equals_pillow_bit_for_bit(h, w):
    """SURVEY.md 8 row a1 on the GPU (round 6): rdx_transform_image -- Resize(512) / CenterCrop(448 | 488) / ToTensor / ExpandChannels of ReportDataset.py:96-106 as
    HIP kernels (Pillow's two fixed-point resampling passes, coefficient tables computed in C doubles by librdx) -- against the host path (PIL.Image.resize, the call
    torchvision's Resize makes) and the numpy restatement: identical float32 tensors, every element. Shapes: both orientations, up-scaling, a skipped pass, no
    resize, a full-size radiograph."""
    import numpy as np
    from PIL import Image
    from oracle import pil_resize as pr
    from radialog_amd import transforms as T
    from radialog_amd.config import small_cfg
    from radialog_amd.engine import RdxEngine
    eng = RdxEngine(small_cfg(), dtype="f16", device=0, max_batch=1, max_len=64, llama=False)       # no weights needed for the transform
    rng = np.random.default_rng(h * 13 + w)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    a[: h // 9] = 0
    a[-2:, :] = 255
    pil = Image.fromarray(a)
    for crop in (448, 488):
        host = T.create_chest_xray_transform_for_inference(512, center_crop_size=crop)(pil)
        dev = T.create_chest_xray_transform_for_inference(512, center_crop_size=crop, engine=eng)(pil)
        assert dev.is_cuda and dev.shape == (3, crop, crop) and dev.dtype == torch.float32
        assert torch.equal(dev.cpu(), host), f"{h}x{w} crop {crop}: {int((dev.cpu() != host).sum())} elements differ from the PIL path"
        assert np.array_equal(dev.cpu().numpy(), pr.inference_transform(a, 512, crop))
    with pytest.raises(Exception, match="smaller than"):
        eng.transform_image(torch.zeros(100, 900, dtype=torch.uint8), 512, 448 + 4096)
    with pytest.raises(ValueError):
        eng.transform_image(torch.zeros(3, 64, 64, dtype=torch.uint8))
    eng.close()

"""Host-side mirror of the reference's Python surface: prompt strings, sharding, tokenizer stand-in, config plumbing,
error behaviour of the generate() wrapper that does not need a GPU."""
import json
import os
import types

import pytest
import torch

from radialog_amd import synth
from radialog_amd.prompter import Conversation, Prompter, SeparatorStyle, new_conversation, report_prompt
from radialog_amd.shard import shard_range
from radialog_amd.tokenizer import IMG_ID, SyntheticTokenizer


def test_prompter_matches_reference_golden(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "prompter.json")))
    pr = Prompter("vicuna_v11")
    for args, expect in gold["generate_prompt"]:
        assert pr.generate_prompt(*args) == expect
    for s, expect in gold["get_response"]:
        assert pr.get_response(s) == expect
    with pytest.raises(ValueError, match="Can't read"):
        Prompter("does_not_exist")


def test_conversation_two_style_prompt():
    conv = new_conversation()
    conv.append_message("USER", "hello")
    conv.append_message("ASSISTANT", None)
    # demo.py:91-99: system + sep, then "ROLE: msg" + seps[i % 2], empty message -> "ROLE:"
    assert conv.get_prompt() == conv.system + " " + "USER: hello " + "ASSISTANT:"
    conv.messages.pop()
    conv.append_message("ASSISTANT", "fine")
    conv.append_message("USER", "next")
    conv.append_message("ASSISTANT", None)
    assert conv.get_prompt() == conv.system + " USER: hello ASSISTANT: fine</s>USER: next ASSISTANT:"
    single = Conversation(system="S", roles=["A", "B"], messages=[["A", "x"], ["B", None]], offset=0)
    assert single.get_prompt() == "S### A: x### B:"
    assert single.sep_style == SeparatorStyle.SINGLE


def test_report_prompt_has_32_image_slots():
    p = report_prompt("edema")
    assert p.count("<IMG>") == 32 and "Predicted Findings: edema." in p
    ids = SyntheticTokenizer()(p)["input_ids"]
    assert int((ids == IMG_ID).sum()) == 32
    # the 32 slots are contiguous (split_at_img relies on it, modeling_llama_imgemb.py:502-503)
    pos = (ids[0] == IMG_ID).nonzero().flatten()
    assert int(pos[-1] - pos[0]) == 31


def test_tokenizer_left_pads_with_unk_zero():
    tok = SyntheticTokenizer()
    out = tok.batch_encode_plus(["a b c d e", "a"], padding=True)
    ids = out["input_ids"]
    assert ids.shape[0] == 2 and ids[1, 0] == 0 and ids[1, -2] == 1          # pads first, BOS right before the text
    assert out["attention_mask"].tolist()[1][:-2] == [0] * (ids.shape[1] - 2)


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_synth_is_deterministic_and_name_keyed():
    a = synth.synth("w.x", (4, 5), -1, 1)
    b = synth.synth("w.x", (4, 5), -1, 1)
    c = synth.synth("w.y", (4, 5), -1, 1)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.min()) >= -1 and float(a.max()) < 1
    ids = synth.synth_prompt_ids(4, 160, pad_rows=True)
    assert ids.shape == (4, 160) and int((ids[0] == 32000).sum()) == 32 and int(ids[0, 0]) == 1
    assert int((ids[3] == 0).sum()) > 0 and int((ids[3] == 32000).sum()) == 32      # left-padded row keeps its slots


def test_config_yaml_and_registry(tmp_path):
    from radialog_amd.blip2_qformer import Blip2Qformer, Config, registry, tasks
    p = tmp_path / "c.yaml"
    p.write_text("model:\n  arch: blip2\n  vit_model: biovil\n  image_size: 448\n  num_query_token: 32\n  synthetic: true\n")
    cfg = Config(types.SimpleNamespace(cfg_path=str(p), options=["model.max_txt_len=95"]))
    assert cfg.model_cfg.arch == "blip2" and cfg.model_cfg.max_txt_len == 95
    assert registry.get_model_class("blip2") is Blip2Qformer
    model = tasks.setup_task(cfg).build_model(cfg)
    assert isinstance(model, Blip2Qformer) and model.cfg.qformer.n_query == 32 and model.eval() is model
    with pytest.raises(KeyError):
        registry.get_model_class("nope")


def test_generate_wrapper_argument_errors():
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM
    lm = LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.float16, synthetic=True)
    ids = torch.ones(1, 40, dtype=torch.long)
    with pytest.raises(NotImplementedError):
        lm.generate(input_ids=ids, do_sample=True, max_new_tokens=4)
    with pytest.raises(ValueError):
        lm.generate(input_ids=ids, num_beams=0, max_new_tokens=4)
    with pytest.raises(KeyError):                                   # unknown dicom id, like the reference's dict lookup
        lm.generate(input_ids=ids, dicom=["unknown"], max_new_tokens=4)
    with pytest.raises(ValueError):
        lm.generate(input_ids=ids, qformer_embs=torch.zeros(1, 31, 768), max_new_tokens=4)
    with pytest.raises(ValueError):
        lm.generate(input_ids=ids[0], max_new_tokens=4)


def test_reusable_prefix_rules_for_multi_turn_kv_reuse():
    """RdxEngine._reusable_prefix (host side of rdx_generate_append): the kept length is the longest token prefix ALL rows share
    with what the cache holds, at least one token is left to run, and nothing is reused across images, batch sizes, or when
    the remainder holds padding or <IMG> tokens."""
    import types
    import torch
    from radialog_amd.engine import RdxEngine
    seq = torch.tensor([[0, 0, 1, 5, 32000, 32000, 7, 8, 9, 11], [1, 4, 32000, 32000, 6, 7, 8, 9, 10, 12]])
    qf = torch.ones(2, 32, 4)
    eng = types.SimpleNamespace(_conv={"seq": seq, "qf": qf.clone()})
    f = lambda ids, q=qf, pad=0: RdxEngine._reusable_prefix(eng, ids, q, pad)
    tail = torch.tensor([[21, 22, 23], [24, 25, 26]])
    assert f(torch.cat([seq, tail], 1)) == 10                    # whole cached sequence kept, the new turn is the tail
    assert f(seq) == 9                                           # identical prompt: one token must still be run
    d = torch.cat([seq, tail], 1)
    d[1, 7] = 99
    assert f(d) == 7                                             # rows must agree: the shortest common prefix over the batch
    assert f(torch.cat([seq, tail], 1), q=qf + 1) == 0           # another image
    assert f(torch.cat([seq, tail], 1), q=None) == 0
    assert f(torch.cat([seq[:1], tail[:1]], 1), q=qf[:1]) == 0   # another batch size
    p = torch.cat([seq, tail], 1)
    p[0, 11] = 0
    assert f(p) == 0                                             # padding behind the kept part
    g = torch.cat([seq, tail], 1)
    g[:, 3] = 77
    assert f(g) == 0                                             # diverges before the <IMG> block ends -> <IMG> in the remainder
    eng._conv = None
    assert f(torch.cat([seq, tail], 1)) == 0


def test_peft_wrapper_forwards_attribute_writes_to_the_wrapped_model():
    """demo.py sets `lang_model.reuse_prefix_kv = True` on whatever init_vicuna returned -- with --lora_model that is the
    PeftModelForCausalLM wrapper, and generate() reads the flag on the inner LlamaForCausalLM."""
    from radialog_amd.config import small_cfg
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM, PeftModelForCausalLM
    inner = LlamaForCausalLM(cfg=small_cfg().llama, dtype="f16", max_batch=1, max_len=64)
    w = PeftModelForCausalLM(inner)
    assert not getattr(inner, "reuse_prefix_kv", False)
    w.reuse_prefix_kv = True
    assert inner.reuse_prefix_kv is True and w.reuse_prefix_kv is True
    assert "reuse_prefix_kv" not in w.__dict__ and w.base_model.model is inner
    w.half().eval()
    assert w.base_model.model is inner


def test_test_py_sizes_the_engine_for_beams_and_downstream_passes():
    """test.py's max_batch arithmetic: rows = batch x beams for the report loop, 14 greedy rows for the binary QA, 5 x
    beams for the findings QA, never more than librdx's 32 rows -- the report loop is chunked instead."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "test.py")
    spec = importlib.util.spec_from_file_location("rdx_test_entry", path)       # (`import test` would find the stdlib's package)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    size = mod.engine_rows
    with pytest.raises(ValueError):
        size(1, 9)
    assert size(12, 1) == (12, 12, 5)
    assert size(12, 3) == (36, 12, 5)                       # round 5: the reference's 12 prompts x 3 beams in one pass (33-128 rows: the row-block family)
    assert size(12, 3, fp8=True) == (36, 12, 5)             # fp8 weights: the same 128 rows since round 5 (the fp8 x fp8 row blocks)
    assert size(12, 3, bin_qa=True, fp8=True) == (36, 12, 5)    # the greedy binary QA needs 14 rows, not 14 x 3
    assert size(2, 1, bin_qa=True) == (14, 2, 5)
    assert size(1, 8, all_qa=True) == (40, 1, 5)            # findings QA: 5 prompts x 8 beams
    assert size(1, 8, all_qa=True, fp8=True) == (40, 1, 5)
    assert size(64, 3) == (126, 42, 5)                      # 192 rows would not fit: chunks of 42 prompts x 3 beams
    for args_ in ((12, 1), (12, 3, True, True), (32, 2), (5, 8, True, True)):
        assert size(*args_)[0] <= 128 and size(*args_, fp8=True)[0] <= 128

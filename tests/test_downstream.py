"""SURVEY.md 8f rank 4 -- the downstream re-prompting callers (test.py:437-674, downstream_tasks/*.py): prompt builders pinned to the
reference's own functions (tests/golden/downstream.json, written by oracle/make_golden.py), the loops' generate arguments and answer
parsing checked on a scripted model (CPU) and, on the GPU, end to end with prefix-KV reuse across the follow-up turns."""
import json
import os

import numpy as np
import pytest
import torch

from radialog_amd import downstream as D


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "downstream.json")))


def test_prompt_builders_match_the_reference(gold):
    cols, P, L = gold["cols"], np.array(gold["preds"]), np.array(gold["labels"])
    assert D.get_correction_prompts(list(gold["history"]), cols, P, L) == gold["correction_prompts"]
    fps, fns = D.get_correction_labels(cols, P, L)
    assert [list(fps), list(fns)] == gold["correction_labels"]
    assert D.get_chexpert_prompts_bin(list(gold["history"]), cols) == gold["bin_prompts"]
    assert D.get_chexpert_prompts_all(list(gold["history"]), cols) == gold["all_prompts"]
    assert "KEEP_OLD" in gold["correction_prompts"][3] and gold["correction_prompts"][1].count("No Finding") == 0


class _Scripted:
    """Stands in for (lang_model, tokenizer): records the generate arguments, answers from a script keyed by the last question."""

    def __init__(self, answers):
        self.answers, self.calls, self._texts = answers, [], None

    def batch_encode_plus(self, texts, return_tensors="pt", padding=True):
        self._texts = list(texts)
        return {"input_ids": torch.zeros(len(texts), 4, dtype=torch.long)}

    def generate(self, input_ids=None, **kw):
        self.calls.append({"batch": input_ids.shape[0], **{k: kw[k] for k in ("max_new_tokens", "dicom", "num_beams")}})
        self._out = [t + " " + next((a for q, a in self.answers if q in t.split("USER:")[-1]), "no.") for t in self._texts]
        return type("O", (), {"sequences": torch.zeros(len(self._texts), 1, dtype=torch.long)})()

    def batch_decode(self, sequences, skip_special_tokens=True):
        return self._out


def test_loops_call_generate_like_test_py_and_parse_answers(gold):
    cols = gold["cols"]
    m = _Scripted([("Cardiomegaly", "Yes, there is."), ("Edema", "yes"), ("List all", "cardiomegaly and pleural effusion."), ("Include", "Updated report.")])
    out = D.run_binary_qa(m, m, [list(q) for q in gold["bin_prompts"][:2]], cols, dicoms=["d0", "d1"])
    assert out.shape == (2, 14) and out[0].tolist() == [0, 0, 1, 0, 0, 1] + [0] * 8          # No Finding derived: 0 (findings present)
    assert all(c["batch"] == 14 and c["max_new_tokens"] == 10 and c["dicom"] == ["d0"] * 14 or c["dicom"] == ["d1"] * 14 for c in m.calls)
    m2 = _Scripted([("Cardiomegaly", "no"), ("Edema", "no")])
    assert D.run_binary_qa(m2, m2, [list(gold["bin_prompts"][0])], cols)[0, 0] == 1             # nothing found -> No Finding
    m.calls.clear()
    allq = D.run_findings_qa(m, m, list(gold["all_prompts"]) * 2, cols, dicoms=[f"d{i}" for i in range(8)])
    assert allq.shape == (8, 14) and allq[0].tolist() == [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0]
    assert [c["batch"] for c in m.calls] == [5, 3] and all(c["max_new_tokens"] == 30 for c in m.calls)
    m.calls.clear()
    corr = D.run_correction(m, m, list(gold["correction_prompts"]), dicoms=["a", "b", "c", "d"], num_beams=1)
    assert len(corr) == 4 and corr[3] == "Stable." and corr[0] == "Updated report."          # KEEP_OLD keeps the first report
    assert [c["batch"] for c in m.calls] == [1, 1, 1] and all(c["max_new_tokens"] == 256 for c in m.calls)


@pytest.mark.gpu
def test_follow_up_turns_reuse_the_prefix_kv_and_match_full_recompute():
    """The report turn, then the correction turn and the findings question on top of it, with `reuse_prefix_kv`: every follow-up call
    must keep the KV rows of the conversation so far and give the ids a fresh engine gives for the same full prompt."""
    from radialog_amd import synth
    from radialog_amd.config import small_cfg
    from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM
    from radialog_amd.prompter import new_conversation, report_prompt
    from radialog_amd.tokenizer import SyntheticTokenizer
    cfg = small_cfg()
    tok = SyntheticTokenizer()
    mk = lambda: LlamaForCausalLM.from_pretrained(None, torch_dtype=torch.float16, cfg=cfg.llama, max_batch=1, max_len=512, synthetic=True).eval()
    lm, fresh = mk(), mk()
    lm.reuse_prefix_kv = True
    emb = synth.synth("t.ds", (32, cfg.llama.qformer_dim), -1.0, 1.0).numpy()
    for m in (lm, fresh):
        m.model.blip_embeddings["dcm"] = emb
    conv = new_conversation()
    conv.append_message(conv.roles[0], report_prompt("cardiomegaly"))
    conv.append_message(conv.roles[1], None)
    ids1 = tok.batch_encode_plus([conv.get_prompt()], return_tensors="pt", padding=True)["input_ids"]
    o1 = lm.generate(input_ids=ids1, dicom=["dcm"], return_dict_in_generate=True, output_scores=True, max_new_tokens=12, eos_token_id=-1)
    assert lm._engine.last_kept_prefix == 0
    # follow-up turn on the TOKEN sequence the model holds (the reference re-tokenises the decoded text; same ids with a real tokenizer)
    follow = tok.batch_encode_plus(["</s>USER: The patient also has edema, correct the report. Don't make other changes. ASSISTANT:"],
                                   return_tensors="pt")["input_ids"][:, 1:]
    ids2 = torch.cat([o1.sequences.cpu(), follow], dim=1)
    o2 = lm.generate(input_ids=ids2, dicom=["dcm"], return_dict_in_generate=True, output_scores=True, max_new_tokens=10, eos_token_id=-1)
    assert lm._engine.last_kept_prefix == o1.sequences.shape[1] - 1            # everything the cache held was kept
    f2 = fresh.generate(input_ids=ids2, dicom=["dcm"], return_dict_in_generate=True, output_scores=True, max_new_tokens=10, eos_token_id=-1)
    assert fresh._engine.last_kept_prefix == 0
    assert torch.equal(o2.sequences, f2.sequences)
    assert max(float((a.float() - b.float()).abs().max()) for a, b in zip(o2.scores, f2.scores)) < 1e-2
    q = tok.batch_encode_plus(["</s>USER: List all the findings in this report. ASSISTANT:"], return_tensors="pt")["input_ids"][:, 1:]
    ids3 = torch.cat([o2.sequences.cpu(), q], dim=1)
    o3 = lm.generate(input_ids=ids3, dicom=["dcm"], return_dict_in_generate=True, max_new_tokens=6, eos_token_id=-1)
    assert lm._engine.last_kept_prefix == o2.sequences.shape[1] - 1
    f3 = fresh.generate(input_ids=ids3, dicom=["dcm"], return_dict_in_generate=True, max_new_tokens=6, eos_token_id=-1)
    assert torch.equal(o3.sequences, f3.sequences)
    lm._engine.close(); fresh._engine.close()

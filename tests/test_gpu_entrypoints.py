"""The entry scripts themselves, executed (VERDICT r3 "missing" 6): demo.py's `main` (mirror of the reference's demo.py:245-305 request
flow) and test.py's `main` (test.py:327-373 report loop, :440-674 downstream passes) on the deterministic random-init weights
(`--synthetic`: no checkpoint is reachable offline) at the REAL sizes -- 448 px encoder, Vicuna-7B decoder -- with short token budgets.
Asserted: what the scripts print / return is what their callees produce for the same inputs (prompt ids with the 32 <IMG> slots, the
hand-off file, generated-id matrices, label matrices), i.e. the plumbing between the mirrors, not the kernels (tests/test_gpu_parity.py)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(f"rdx_entry_{name}", os.path.join(REPO, f"{name}.py"))   # `import test` = the stdlib's package
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("classifier", [False, True])
def test_demo_main_runs_one_request(tmp_path, monkeypatch, capsys, classifier):
    """demo.py:245-305: image -> forward_image -> current_chat_img.pt -> prompt with 32 x <IMG> -> generate -> text after ASSISTANT:."""
    from radialog_amd.chexpert_model import CHEXPERT_COLS
    from radialog_amd.prompter import new_conversation, report_prompt
    from radialog_amd.tokenizer import IMG_ID, load_tokenizer
    monkeypatch.chdir(tmp_path)
    demo = _load("demo")
    N = 8
    argv = ["--synthetic", "--max_new_tokens", str(N), "--dtype", "f16"] + ([] if classifier else ["--no-classifier"])
    res = demo.main(argv)
    out = capsys.readouterr().out
    # the hand-off file of the reference (demo.py:273) was written and holds the Q-Former output of ONE image
    qf = torch.load(tmp_path / "current_chat_img.pt")
    assert tuple(qf.shape) == (1, 32, 768) and qf.dtype == torch.float32 and torch.isfinite(qf).all()
    # findings: the classifier's label names (demo.py:256-261) or the fixed text
    if classifier:
        assert "predicted findings:" in out
        names = {c.lower() for c in CHEXPERT_COLS}                      # demo.py:257-262: ', '.join(class names).lower()
        assert res["findings"] == "no finding" or all(f.strip() in names for f in res["findings"].split(","))
    else:
        assert res["findings"] == "no finding"
    # sequences = prompt ids (with exactly 32 <IMG> slots) + N generated ids (random weights never emit EOS within 8 tokens)
    conv = new_conversation()
    conv.append_message(conv.roles[0], report_prompt(res["findings"]))
    conv.append_message(conv.roles[1], None)
    want = load_tokenizer(None)(conv.get_prompt())["input_ids"]
    seq = res["sequences"]
    assert seq.shape[0] == 1 and seq.shape[1] == want.shape[1] + N and res["n_scores"] == N
    assert torch.equal(seq[:, : want.shape[1]].long(), want.long())
    assert int((seq == IMG_ID).sum()) == 32
    assert f"generated {seq.shape[1]} ids, {N} steps" in out and "ASSISTANT:" in out
    assert isinstance(res["prediction"], str) and len(res["prediction"].split()) >= 1


def test_demo_main_on_an_image_file_runs_the_gpu_transform(tmp_path, monkeypatch, capsys):
    """`demo.py --image x.png`: load_image / remap_to_uint8 (demo.py:173-218) on the host, then Resize(512) / CenterCrop(448 | 488) / ToTensor / ExpandChannels on the
    GPU (rdx_transform_image through the decoder's context) for the report model AND the findings classifier; the tensors equal the host (PIL) path's."""
    import numpy as np
    from PIL import Image
    monkeypatch.chdir(tmp_path)
    demo = _load("demo")
    rng = np.random.default_rng(11)
    raw = (rng.random((760, 640)) * 3000 + 50).astype(np.uint16)
    Image.fromarray(raw).save(tmp_path / "cxr.png")
    res = demo.main(["--synthetic", "--max_new_tokens", "4", "--dtype", "f16", "--image", str(tmp_path / "cxr.png")])
    out = capsys.readouterr().out
    assert "predicted findings:" in out and res["n_scores"] == 4 and res["sequences"].shape[0] == 1
    from radialog_amd.config import small_cfg
    from radialog_amd.engine import RdxEngine
    eng = RdxEngine(small_cfg(), dtype="f16", device=0, max_batch=1, max_len=64, llama=False)
    for crop in (448, 488):
        host = demo.load_image(str(tmp_path / "cxr.png"), crop=crop)
        dev = demo.load_image(str(tmp_path / "cxr.png"), crop=crop, engine=eng)
        assert dev.is_cuda and torch.equal(dev.cpu(), host)
    eng.close()


def test_demo_generate_equals_the_engine_on_the_same_inputs(tmp_path, monkeypatch):
    """The mirrors add nothing and lose nothing: demo.main's generated ids == RdxEngine.generate on the prompt ids and the Q-Former
    output read back from the hand-off file."""
    from radialog_amd.config import full_cfg
    from radialog_amd.engine import RdxEngine, synth_getter
    monkeypatch.chdir(tmp_path)
    demo = _load("demo")
    N = 6
    res = demo.main(["--synthetic", "--no-classifier", "--max_new_tokens", str(N), "--dtype", "f16"])
    seq = res["sequences"]
    T = seq.shape[1] - N
    qf = torch.load(tmp_path / "current_chat_img.pt")
    cfg = full_cfg()
    eng = RdxEngine(cfg, dtype="f16", device=0, max_batch=1, max_len=256, lora=True, vision=False)
    eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
    toks, _, n = eng.generate(seq[:, :T].long(), qf, max_new=N, eos_id=2, pad_id=0)
    eng.close()
    assert toks[0, :N].cpu().long().tolist() == seq[0, T:].long().tolist()


def test_test_py_report_loop_embeddings_beams_and_downstream(capsys):
    """test.py:327-373 + :440-674: 5 studies in batches of 3 (left-padded batch + a ragged last batch), greedy; the same with the
    {dicom: float32[32,768]} embeddings (--use_embs) -> other tokens; --num_beams 3; the three downstream passes."""
    t = _load("test")
    base = ["--synthetic", "--num_samples", "5", "--batch_size", "3", "--max_new_tokens", "6", "--dtype", "f16"]
    r0 = t.main(base)
    out = capsys.readouterr().out
    assert "generated 5 reports x 6 token slots on 1 GPU(s)" in out
    assert tuple(r0["ids"].shape) == (5, 6) and len(r0["preds"]) == 5 and int((r0["ids"] != 0).sum()) >= 25
    # identical prompts without image rows -> identical reports for the studies of one batch (batches of 3 and 2 run different kernels)
    assert torch.equal(r0["ids"][0], r0["ids"][1]) and torch.equal(r0["ids"][0], r0["ids"][2]) and torch.equal(r0["ids"][3], r0["ids"][4])

    r1 = t.main(base + ["--use_embs"])
    capsys.readouterr()
    assert tuple(r1["ids"].shape) == (5, 6)
    # every study now carries its own image embedding: the reports differ from the no-image ones and from each other
    assert not torch.equal(r1["ids"], r0["ids"])
    assert len({tuple(r.tolist()) for r in r1["ids"]}) > 1

    r2 = t.main(base + ["--use_embs", "--num_beams", "3"])
    capsys.readouterr()
    assert tuple(r2["ids"].shape) == (5, 6) and r2["batch_size"] == 3

    r3 = t.main(["--synthetic", "--num_samples", "2", "--batch_size", "2", "--max_new_tokens", "6", "--dtype", "f16", "--use_embs",
                 "--do_corr", "--do_cp_bin_qa", "--do_cp_all_qa"])
    out = capsys.readouterr().out
    assert len(r3["corrected"]) == 2 and all(isinstance(c, str) for c in r3["corrected"])
    assert r3["bin_qa"].shape == (2, 14) and r3["all_qa"].shape == (2, 14)
    assert set(np.unique(r3["bin_qa"])) <= {0, 1} and set(np.unique(r3["all_qa"])) <= {0, 1}
    assert "corrected reports" in out and "binary QA label matrix" in out and "findings QA label matrix" in out

#!/usr/bin/env python
"""test.py -- the reference's batched report-generation loop (test.py:277-373) on the MI355X-native path.

Same flags (`--prompt`, `--lora_model`, `--num_workers`, `--use_embs`, `--num_beams`); batches of 12 left-padded
prompts, `lang_model.generate(input_ids=..., dicom=dicom_id if use_embs else None, max_new_tokens=300,
return_dict_in_generate=True, output_scores=True)`, predictions = text after "ASSISTANT:". MIMIC-CXR is credentialed
data and the NLG/CheXbert metrics need Java and a second conda env, so the dataset is replaced by synthetic studies
(`--num_samples`) whose image embeddings are produced by the encoder half first (the reference's
pretraining/train.py evaluate branch) and handed over through the same {dicom: float32[32,768]} mapping.
With WORLD_SIZE > 1 (torch.distributed.run) the studies are sharded across ranks and predictions all-gathered."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from radialog_amd import synth                                              # noqa: E402
from radialog_amd.embed_dump import dump_embeddings                         # noqa: E402
from radialog_amd.modeling_llama_imgemb import LlamaForCausalLM, PeftModelForCausalLM   # noqa: E402
from radialog_amd.prompter import new_conversation, report_prompt          # noqa: E402
from radialog_amd.shard import allgather_ragged, shard_range               # noqa: E402
from radialog_amd.tokenizer import load_tokenizer                          # noqa: E402


ENGINE_ROWS = 128     # librdx holds up to 128 decoder rows per context in the model dtype (rdx_ctx.h RDX_MAX_ROWS; round 5: the row-block decode family) ...
ENGINE_ROWS_FP8 = 128  # ... and with fp8 weights (round 5: the 32-row fp8 x fp8 decode kernels per row block)


def engine_rows(batch_size, num_beams, bin_qa=False, all_qa=False, fp8=False):
    """Rows the engine must hold at once -> (max_batch, report-loop batch size, findings-QA batch size): the report loop runs batch_size
    prompts x num_beams beam rows -- the reference's 12 x 3 = 36 (test.py:267,:279) in ONE pass since round 5 (chunked only when that exceeds the
    engine's rows: 128); the binary QA pass is greedy (14 questions per study, test.py:548-590); the findings QA runs
    batches of 5 with beams (test.py:610-650)."""
    rows = ENGINE_ROWS_FP8 if fp8 else ENGINE_ROWS
    beams = max(num_beams, 1)
    if beams > 8:
        raise ValueError(f"--num_beams {beams}: rdx_beam_search supports at most 8 beams")
    if batch_size * beams > rows:
        chunk = max(1, rows // beams)
        print(f"note: --batch_size {batch_size} x --num_beams {beams} exceeds the engine's {rows} rows; the report loop runs in chunks of {chunk}")
        batch_size = chunk
    qa_batch = max(1, min(5, rows // beams))
    max_batch = max(batch_size * beams, 14 if bin_qa else 0, qa_batch * beams if all_qa else 0, beams, 1)
    return max_batch, batch_size, qa_batch


def main(argv=None):
    """Returns {"ids", "preds", ...} (the reference's script only prints; the return value is what tests/test_gpu_entrypoints.py asserts on)."""
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, default="img_matching_examples_ig2_noexamples_IMG_findings")
    p.add_argument("--lora_model", type=str, default=None)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--use_embs", action="store_true", default=False)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--vicuna", default=None)
    p.add_argument("--num_samples", type=int, default=24)
    p.add_argument("--batch_size", type=int, default=12)
    p.add_argument("--max_new_tokens", type=int, default=300)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    p.add_argument("--synthetic", action="store_true", help="deterministic random-init weights (no checkpoints reachable offline)")
    p.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: decoder GEMMs in OCP e4m3 on the fp8 MFMA (weights_fp8=True)")
    p.add_argument("--do_corr", action="store_true", help="automatic prompt correction on top of the reports (test.py:437-500)")
    p.add_argument("--do_cp_bin_qa", action="store_true", help="14 yes/no CheXpert questions per study (test.py:545-590)")
    p.add_argument("--do_cp_all_qa", action="store_true", help="'List all the findings' follow-up (test.py:608-650)")
    args = p.parse_args(argv)
    res = {}

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    lo, hi = shard_range(args.num_samples, world, rank)
    dicoms = [f"synthetic-{i:05d}" for i in range(lo, hi)]
    tok = load_tokenizer(args.vicuna)
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    try:
        max_batch, args.batch_size, qa_batch = engine_rows(args.batch_size, args.num_beams, args.do_cp_bin_qa, args.do_cp_all_qa, fp8=args.fp8)
    except ValueError as e:
        p.error(str(e))
    beams = max(args.num_beams, 1)
    lang_model = LlamaForCausalLM.from_pretrained(args.vicuna, torch_dtype=dt, device_map="auto", max_batch=max_batch,
                                                  max_len=1024, device=local, synthetic=args.synthetic, weights_fp8=args.fp8)
    if args.lora_model:
        lang_model = PeftModelForCausalLM.from_pretrained(lang_model, args.lora_model, torch_dtype=dt, use_ram_optimized_load=False).half()
    lang_model.eval()
    if args.use_embs:
        images = synth.synth_images(len(dicoms), 448, seed=1000 + lo)
        lang_model.model.blip_embeddings.update(dump_embeddings(images, dicoms, dtype=args.dtype, device=local, synthetic=args.synthetic))

    all_preds, all_ids, preds_history = [], [], []
    for s in range(0, len(dicoms), args.batch_size):
        batch = dicoms[s: s + args.batch_size]
        texts = []
        for d in batch:
            conv = new_conversation()
            conv.append_message(conv.roles[0], report_prompt("no finding"))
            conv.append_message(conv.roles[1], None)
            texts.append(conv.get_prompt())
        input_ids = tok.batch_encode_plus(texts, return_tensors="pt", padding=True)["input_ids"]
        out = lang_model.generate(input_ids=input_ids, dicom=batch if args.use_embs else None, return_dict_in_generate=True,
                                  output_scores=True, max_new_tokens=args.max_new_tokens, num_beams=args.num_beams)
        preds = tok.batch_decode(out.sequences, skip_special_tokens=True)
        preds_history.extend(preds)
        all_preds.extend([q.split("ASSISTANT:")[1] if "ASSISTANT:" in q else q for q in preds])
        gen = out.sequences[:, input_ids.shape[1]:]
        pad = torch.zeros(gen.shape[0], args.max_new_tokens, dtype=torch.int32, device=gen.device)
        pad[:, : gen.shape[1]] = gen.to(torch.int32)
        all_ids.append(pad)
    ids = torch.cat(all_ids, 0)
    if world > 1:
        counts = [shard_range(args.num_samples, world, r)[1] - shard_range(args.num_samples, world, r)[0] for r in range(world)]
        ids = allgather_ragged(ids, counts, world)
    if args.do_corr or args.do_cp_bin_qa or args.do_cp_all_qa:
        # downstream re-prompting (test.py:437-674): the whole conversation + one new USER turn; CheXbert labels are not available here,
        # so the correction prompts are driven by synthetic label disagreements of the same shape
        from radialog_amd import downstream as D
        from radialog_amd.chexpert_model import CHEXPERT_COLS
        import numpy as np
        rng = np.random.default_rng(0)
        dc = dicoms if args.use_embs else None
        if args.do_corr:
            P, L = rng.integers(0, 2, (len(dicoms), 14)), rng.integers(0, 2, (len(dicoms), 14))
            corr = D.run_correction(lang_model, tok, D.get_correction_prompts(list(preds_history), CHEXPERT_COLS, P, L), dc, args.num_beams)
            res["corrected"] = corr
            print(f"rank {rank}: {len(corr)} corrected reports; first: {corr[0][:80]!r}")
        if args.do_cp_bin_qa:
            yn = D.run_binary_qa(lang_model, tok, D.get_chexpert_prompts_bin(list(preds_history), CHEXPERT_COLS), CHEXPERT_COLS, dc)
            res["bin_qa"] = yn
            print(f"rank {rank}: binary QA label matrix {yn.shape}, positives {int(yn.sum())}")
        if args.do_cp_all_qa:
            oh = D.run_findings_qa(lang_model, tok, D.get_chexpert_prompts_all(list(preds_history), CHEXPERT_COLS), CHEXPERT_COLS, dc,
                                   batch_size=qa_batch, num_beams=args.num_beams)
            res["all_qa"] = oh
            print(f"rank {rank}: findings QA label matrix {oh.shape}, positives {int(oh.sum())}")
    if rank == 0:
        print(f"generated {ids.shape[0]} reports x {ids.shape[1]} token slots on {world} GPU(s); first: {all_preds[0][:120]!r}")
    res.update({"ids": ids.cpu(), "preds": all_preds, "dicoms": dicoms, "batch_size": args.batch_size})
    lang_model.close()
    return res


if __name__ == "__main__":
    main()

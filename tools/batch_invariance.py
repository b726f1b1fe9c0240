"""Full-depth sanity of the batch paths (32 layers, production widths): row 0 of the same prompts decoded at batch 32, 4 and 1 --
finite logits, identical greedy tokens in fp16 (bf16 rounding noise over 32 layers of random-init weights flips near-ties):
python tools/batch_invariance.py [f16|bf16]"""
import sys
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
cfg = full_cfg()
T, N = 160, 12
res = {}
for B in (32, 4, 1):
    eng = RdxEngine(cfg, dtype=(sys.argv[1] if len(sys.argv) > 1 else "f16"), device=0, max_batch=B, max_len=256, lora=True, vision=False)
    eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
    ids = synth.synth_prompt_ids(32, T, vocab=cfg.llama.vocab, pad_rows=False, seed=7)[:B].to(eng.device)
    qf = synth.synth("u.qf", (32, 32, cfg.llama.qformer_dim), -1.0, 1.0)[:B].to(eng.device)
    toks, scores, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0, output_scores=True, use_graph=True)
    res[B] = (toks.cpu().clone(), scores.float().cpu().clone())
    print(B, "finite", bool(torch.isfinite(scores.float()).all()), "tokens row0", toks[0].tolist())
    eng.close()
for B in (32, 4):
    t, s = res[B]; t1, s1 = res[1]
    same = (t[0] == t1[0]).long().cumprod(0).sum().item()
    print(f"B={B} vs B=1 row 0: identical tokens for the first {same}/{N} steps; step-0 logits max diff {float((s[0,0]-s1[0,0]).abs().max()):.4f}, step-1 {float((s[1,0]-s1[1,0]).abs().max()):.4f}")

"""Per-step logit error of the production-width batch-20 case against the oracle: python tests/diag/b20_diag.py [dtype]"""
import sys, torch
from radialog_amd import synth
from radialog_amd.config import LlamaCfg, RaDialogCfg
from radialog_amd.engine import RdxEngine, synth_getter
from oracle import ref_cpu
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DT = {"f16": torch.float16, "bf16": torch.bfloat16}
cfg = RaDialogCfg(llama=LlamaCfg(layers=1, qformer_dim=192))
w = synth.make_weights(synth.llama_specs(cfg.llama, lora=True))
B, T, N = 20, 96, 5
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=6, pad_rows=True, seed=5)
qf = synth.synth("t.qf20", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
with torch.no_grad():
    ref = ref_cpu.LlamaOracle(w, cfg.llama, DT[dtype], lora=True).generate_greedy(ids, qf, max_new=N, eos_id=-1, pad_id=0)
eng = RdxEngine(cfg, dtype=dtype, device=0, max_batch=B, max_len=128, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
toks, scores, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0, output_scores=True, use_graph=True)
toks = toks.cpu().long()
for s in range(N):
    r = torch.stack([ref["scores"][s][b].float() for b in range(B)])
    g = scores[s].float().cpu()
    err = (g - r).abs().amax(dim=1)
    same = (toks[:, s] == ref["tokens"][:, s])
    print(f"step {s}: max|logit| {float(r.abs().max()):.2f}  max err {float(err.max()):.4f}  mean err {float((g - r).abs().mean()):.5f}  tokens equal {int(same.sum())}/{B}  min margin among mismatches "
          f"{float(ref['margins'][s][~same].min()) if (~same).any() else -1:.3f}")
for s in range(N):
    g = scores[s].float().cpu()
    r = torch.stack([ref["scores"][s][b].float() for b in range(B)])
    print(f"step {s}: gpu nan rows {torch.isnan(g).any(dim=1).nonzero().flatten().tolist()}  oracle nan rows {torch.isnan(r).any(dim=1).nonzero().flatten().tolist()}")
print("pad lens", (ids == 0).sum(dim=1).tolist())

"""Where do NaNs first appear at production width, batch 20? eager steps, per-step checks: python tests/diag/b20_nan.py [dtype] [B] [max_len]"""
import sys, torch
from radialog_amd import synth
from radialog_amd.config import LlamaCfg, RaDialogCfg
from radialog_amd.engine import RdxEngine, synth_getter
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ML = int(sys.argv[3]) if len(sys.argv) > 3 else 128
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = RaDialogCfg(llama=LlamaCfg(layers=NL, qformer_dim=192))
T, N = 96, 6
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=6, pad_rows=True, seed=5)
qf = synth.synth("t.qf20", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
eng = RdxEngine(cfg, dtype=dtype, device=0, max_batch=B, max_len=ML, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
toks, lg = eng.prefill(ids, qf, max_new=N, eos_id=-1, pad_id=0)
print("prefill: nan rows", torch.isnan(lg.float()).any(dim=1).sum().item(), "max", float(lg.float().abs().max()))
for s in range(1, N):
    toks, lg = eng.decode_step()
    h = eng.hidden_read() if hasattr(eng, "hidden_read") else None
    k = eng.kv_read(0, 0, B).float(); v = eng.kv_read(0, 1, B).float()
    print(f"step {s}: logits nan rows {torch.isnan(lg.float()).any(dim=1).sum().item()} max {float(lg.float().abs().nan_to_num(0).max()):.2f}  "
          f"K nan {int(torch.isnan(k).sum())} inf {int(torch.isinf(k).sum())} max {float(k.abs().nan_to_num(0).max()):.1f}  V nan {int(torch.isnan(v).sum())} max {float(v.abs().nan_to_num(0).max()):.1f}")

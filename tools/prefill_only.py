"""Prefill only (for rocprofv3): python tools/prefill_only.py [B] [T] [iters] [fp8]  -- rdx_generate with max_new = 1"""
import sys, time, torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 160
it = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fp8 = len(sys.argv) > 4 and sys.argv[4] in ("1", "fp8")
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=(T + 63) // 32 * 32, lora=True, vision=False, weights_fp8=fp8)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=(B > 1), seed=7).to(eng.device)
qf = synth.synth("u.qf", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=1, eos_id=-1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(it): eng.generate(ids, qf, max_new=1, eos_id=-1)
torch.cuda.synchronize()
print("prefill B=%d T=%d%s: %.3f ms" % (B, T, " fp8" if fp8 else "", (time.perf_counter() - t0) / it * 1e3))
eng.close()

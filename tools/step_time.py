"""Decode-step time only (hipGraph replay through rdx_time(0)) at full size: python tools/step_time.py [B] [T] [N]"""
import sys
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 160
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=512, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=(B > 1), seed=7).to(eng.device)
qf = synth.synth("u.qf", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
toks, _, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0, use_graph=True)
print("step graph %.1f us   tokens[:8]=%s" % (eng.time_unit(0, 30) * 1e3, toks[0, :8].tolist()))
eng.close()

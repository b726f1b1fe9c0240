"""Per-unit decode timings (HIP events through rdx_time) at full size: python tools/unit_times.py [B] [dtype]"""
import sys
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 160
N = int(sys.argv[4]) if len(sys.argv) > 4 else 128
cfg = full_cfg()
eng = RdxEngine(cfg, dtype=dt, device=0, max_batch=B, max_len=512, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=(B > 1), seed=7).to(eng.device)
qf = synth.synth("u.qf", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0, use_graph=True)
names = {0: "step graph", 1: "gate/up", 2: "qkv", 3: "o_proj", 4: "down", 5: "lm_head", 6: "attention"}
print(f"B={B} T={T} N={N}")
for w, n in names.items():
    ms = eng.time_unit(w, 20 if w else 30)
    print(f"{n:12s} {ms*1e3:9.2f} us")
for layer in (3, 17, 17):
    t = eng.attn_trace(layer).tolist()
    e = t[7]
    print("attn L%d: entry->slot %.2f, ->ready %.2f, ->newtok %.2f, ->sync1 %.2f, ->scores %.2f, ->softmax %.2f, ->pv %.2f us" % (
        layer, *[(t[i] - e) / 100 for i in range(7)]))
eng.close()

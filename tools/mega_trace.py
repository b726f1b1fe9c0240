"""Timeline of one chained decode step (RDX_MEGA=-1): per role and layer, when workgroups start, get inputs, end."""
import os, sys
os.environ.setdefault("RDX_MEGA", "-1")
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter

cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=1, max_len=512, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(1, 160, vocab=cfg.llama.vocab, pad_rows=False, seed=7).to(eng.device)
qf = synth.synth("u.qf", (1, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=128, eos_id=-1, pad_id=0, use_graph=True)
nwg = [193, 32, 256, 344, 256]
per = sum(nwg)
eng.mega_trace(per * 32 + 64)
full = eng.mega_trace(per * 32 + 64).numpy()
tr = full[: per * 32].copy()
att = full[per * 32:].reshape(-1)[: 32 * 8].reshape(32, 8)
t0 = tr[:, 0].min()
tr[:, :3] -= t0
names = ["qkv", "att", "o", "gu", "down"]
print("ticks = 10 ns; columns: first start, last start | first ready, last ready | first end, last end | mean(end-ready)")
for l in (0, 1, 15, 31):
    base = l * per
    off = 0
    for r, n in enumerate(nwg):
        w = tr[base + off: base + off + n]
        off += n
        assert (w[:, 3] == r).all(), (l, r, w[:4])
        print(f"L{l:02d} {names[r]:5s} start {w[:,0].min()/100:8.2f} {w[:,0].max()/100:8.2f} | ready {w[:,1].min()/100:8.2f} {w[:,1].max()/100:8.2f} | end {w[:,2].min()/100:8.2f} {w[:,2].max()/100:8.2f} | work {(w[:,2]-w[:,1]).mean()/100:6.2f} us")
for l in (1, 15):
    a = att[l] - t0
    print(f"L{l:02d} att(h0) start {a[0]/100:.2f} ready {a[1]/100:.2f} newtok {a[2]/100:.2f} sync1 {a[3]/100:.2f} scores+sync {a[4]/100:.2f} softmax {a[5]/100:.2f} pv+sync {a[6]/100:.2f}")
print("total", (tr[:, 2].max()) / 100, "us")
eng.close()

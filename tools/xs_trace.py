"""Timeline inside the batch-32 activation-stationary GEMM (xstat32.hip): python tools/xs_trace.py [what ...]  (1 gate/up, 2 qkv)"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import sys
import numpy as np, torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
cfg = full_cfg()
B = 32
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=512, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(B, 160, vocab=cfg.llama.vocab, pad_rows=True, seed=7).to(eng.device)
qf = synth.synth("u.qf", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=16, eos_id=-1, pad_id=0, use_graph=True)
sel = [int(x) for x in sys.argv[1:]] or [1, 2]
for what, name in ((1, "gate/up"), (2, "qkv")):
    if what not in sel: continue
    for layer in (5, 6):
        raw = eng.gemv_trace(what, layer).numpy()
        raw = raw[raw[:, 0] > 0]
        t0 = raw[:, 0].min()
        tr = (raw[:, :6].astype(np.float64) - t0) / 100.0
        nit = raw[:, 6]
        lab = ["entry", "trip0 K done", "trip0 done", "last trip begins", "last K done", "end"]
        print(f"{name} L{layer}: {len(raw)} workgroups, trips {np.bincount(nit.astype(int)).tolist()}")
        print("   " + " | ".join(f"{lab[i]} {tr[:, i].min():.2f}..{tr[:, i].max():.2f} (med {np.median(tr[:, i]):.2f})" for i in range(6)))
        for k in sorted(set(nit.tolist())):
            m = nit == k
            print(f"   {int(k)} trips: end med {np.median(tr[m, 5]):.2f} max {tr[m, 5].max():.2f}; per-trip (trip0 done -> last begins) {np.median((tr[m, 3] - tr[m, 2]) / max(k - 2, 1)):.2f} us")
eng.close()

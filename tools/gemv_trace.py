"""Timeline inside a stand-alone decode GEMV (batch 1): python tools/gemv_trace.py"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import numpy as np, torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=1, max_len=512, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(1, 160, vocab=cfg.llama.vocab, pad_rows=False, seed=7).to(eng.device)
qf = synth.synth("u.qf", (1, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=16, eos_id=-1, pad_id=0, use_graph=True)
import sys
sel = [int(x) for x in sys.argv[1:]] or [1, 2, 4]
for what, name, nt in ((1, "gate/up", 1376), (2, "qkv", 769), (4, "down", 256)):
    if what not in sel: continue
    for layer in (5, 6):
        raw = eng.gemv_trace(what, layer).numpy()[:nt]
        hw, xcc = raw[:, 6], raw[:, 7] & 0xf
        cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)    # cu_id | sh_id | se_id | xcc
        ids, cnt = np.unique(cu, return_counts=True)
        endt = (raw[:, 5] - raw[:, 0].min()) / 100.0
        print(f"   distinct CUs {len(ids)}, workgroups per CU: min {cnt.min()} max {cnt.max()} hist {np.bincount(cnt).tolist()}")
        for k in sorted(set(cnt.tolist())):
            sel_ = np.isin(cu, ids[cnt == k])
            print(f"      CUs with {k} workgroups: mean end {endt[sel_].mean():.2f} max {endt[sel_].max():.2f} us")
        print("      per-XCD mean/max end: " + " ".join(f"{int(x)}:{endt[xcc == x].mean():.1f}/{endt[xcc == x].max():.1f}" for x in sorted(set(xcc.tolist()))))
        # tile index -> end time trend (tiles are laid out contiguously in HBM)
        q = np.array_split(np.arange(nt), 8)
        print("      by tile octile (mean end): " + " ".join(f"{endt[i].mean():.1f}" for i in q))
        tr = raw.astype(np.float64)
        t0 = tr[:, 0].min()
        tr = (tr[:, :6] - t0) / 100.0
        lab = ["entry", "w issued", "x staged", "K loop done", "waves done", "end"]
        e = np.sort(tr[:, 5]); print(f"   end-time deciles: " + " ".join(f"{e[int(q * (len(e) - 1))]:.1f}" for q in (0, .1, .25, .5, .75, .9, .99, 1)))
        print(f"{name} L{layer}: " + " | ".join(f"{lab[i]} {tr[:, i].min():.2f}..{tr[:, i].max():.2f} (med {np.median(tr[:, i]):.2f})" for i in range(6)))
eng.close()

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", max_batch=1, max_len=448)
eng.load_weights(synth_getter(cfg, eng.device))
ids = synth.synth_prompt_ids(1, 160)
for g in (False, True):
    for n in (64, 200, 240, 256):
        t0 = time.time()
        try:
            toks, _, k = eng.generate(ids, None, max_new=n, eos_id=-1, use_graph=g)
            print("graph" if g else "eager", n, "ok", f"{1e3*(time.time()-t0):.1f} ms", toks[0, :6].tolist(), flush=True)
        except Exception as e:
            print("graph" if g else "eager", n, "ERR", f"{1e3*(time.time()-t0):.1f} ms", str(e)[:120], flush=True)

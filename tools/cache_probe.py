"""Does cache residency (L2 / Infinity Cache) speed up the decode GEMVs?  same-layer (resident) vs layer-sweep (HBM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", max_batch=1, max_len=448)
eng.load_weights(synth_getter(cfg, eng.device))
ids = synth.synth_prompt_ids(1, 160)
eng.generate(ids, None, max_new=8, eos_id=-1)
H, I = 4096, 11008
for what, name, mb in ((1, "gate/up", 2 * I * H * 2 / 1e6), (2, "qkv", 12304 * H * 2 / 1e6), (3, "o_proj", H * H * 2 / 1e6), (4, "down", H * I * 2 / 1e6)):
    a = eng.time_unit(what, 5) * 1e3
    b = eng.time_unit(what + 10, 5) * 1e3
    print(f"{name:8s} {mb:7.1f} MB  sweep {a:6.1f} us ({mb/a/1e3*1e3:5.2f} TB/s)   same-layer {b:6.1f} us ({mb/b:5.2f} TB/s)")

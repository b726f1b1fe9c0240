"""Decode attention alone at batch B over context lengths (rdx_time unit 6): python tools/attn_time.py [B]"""
import sys
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=512, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
for T in (96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448):
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=False, seed=7).to(eng.device)
    eng.generate(ids, None, max_new=2, eos_id=-1, pad_id=0, use_graph=False)
    us = eng.time_unit(6, 5) * 1e3
    by = B * 32 * (T + 1) * 512
    print(f"T={T:4d}: {us:6.2f} us/layer   KV bytes {by/1e6:6.1f} MB -> {by/us/1e6:5.2f} TB/s   floor(6.3 TB/s) {by/6.3e6:5.1f} us")
eng.close()

import sys, os, time
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
ML = int(sys.argv[1]); N = int(sys.argv[2]); G = int(sys.argv[3]); mode = sys.argv[4] if len(sys.argv) > 4 else ""
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=1, max_len=ML, lora=True)
eng.load_weights(synth_getter(cfg, eng.device, lora=True))
img = synth.synth_images(1, cfg.vision.img, seed=16).to(eng.device)
ids = synth.synth_prompt_ids(1, 160, vocab=cfg.llama.vocab, pad_rows=False, seed=7).to(eng.device)
for i in range(G):
    t0 = time.perf_counter()
    q, _ = eng.encode_image(img, want_image_embeds=False)
    toks, _, n = eng.generate(ids, q, max_new=N, eos_id=-1, pad_id=0, use_graph=True)
    torch.cuda.synchronize()
    print("generate %d: %.1f ms, ptr %x" % (i, (time.perf_counter() - t0) * 1e3, toks.data_ptr()))
if "e" in mode:
    for _ in range(5): eng.encode_image(img, want_image_embeds=False)
if "g" in mode: print("gate/up", eng.time_unit(1, 10) * 1e3)
for it in (1, 1, 20):
    t0 = time.perf_counter()
    ms = eng.time_unit(0, it)
    print("iters %d: step graph %.1f us (wall %.1f ms)" % (it, ms * 1e3, (time.perf_counter() - t0) * 1e3))
try:
    toks, _, n = eng.generate(ids, q, max_new=N, eos_id=-1, pad_id=0, use_graph=True)
    print("next generate ok")
except Exception as e:
    print("next generate failed:", e)
eng.close()

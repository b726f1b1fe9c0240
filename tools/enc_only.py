"""Batched encode only (for rocprofv3): python tools/enc_only.py [B] [iters]"""
import sys, time, torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=64, lora=False, llama=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=False), llama=False)
img = synth.synth_images(B, cfg.vision.img, seed=16).to(eng.device)
eng.encode_image(img, want_image_embeds=False); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(it): eng.encode_image(img, want_image_embeds=False)
torch.cuda.synchronize()
print("encode B=%d: %.3f ms/img" % (B, (time.perf_counter() - t0) / it / B * 1e3))
eng.close()

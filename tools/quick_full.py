"""First full-size timing (GPU box): Vicuna-7B shapes, random init, B from argv."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = full_cfg()
t0 = time.time()
eng = RdxEngine(cfg, dtype=dt, max_batch=B, max_len=448)
eng.load_weights(synth_getter(cfg, eng.device))
print("load s", time.time() - t0, "mem GB", torch.cuda.mem_get_info()[0] / 1e9, flush=True)
img = synth.synth_images(B, 448).cuda()
ids = synth.synth_prompt_ids(B, 160, pad_rows=True)
for it in range(3):
    t0 = time.time(); q, _ = eng.encode_image(img, want_image_embeds=False); t1 = time.time()
    toks, _, n = eng.generate(ids, q, max_new=256, eos_id=-1, use_graph=True); t2 = time.time()
    print(f"iter {it}: encode {1e3*(t1-t0):.2f} ms, generate(256) {1e3*(t2-t1):.1f} ms -> {(256*B)/(t2-t1):.1f} tok/s, reports/s {B/(t2-t0):.3f}", flush=True)
t0 = time.time(); toks, _, n = eng.generate(ids, q, max_new=1, eos_id=-1); print("prefill+1 ms", 1e3 * (time.time() - t0))
t0 = time.time(); toks, _, n = eng.generate(ids, q, max_new=256, eos_id=-1, use_graph=False); print("eager generate ms", 1e3 * (time.time() - t0))
print("tokens", toks[0, :8].tolist(), "finite", bool(torch.isfinite(q).all()))
toks, _, n = eng.generate(ids, q, max_new=64, eos_id=-1)
H, I, V = 4096, 11008, 32001
for what, name, byts in ((0, "decode step graph", 13.21e9), (1, "gate/up gemv", 2 * I * H * 2), (2, "qkv gemv", (3 * H + 16) * H * 2),
                         (3, "o_proj gemv", H * H * 2), (4, "down gemv", H * I * 2), (5, "lm_head", V * H * 2)):
    ms = eng.time_unit(what, 20 if what == 0 else 5)
    print(f"{name}: {ms*1e3:.1f} us  -> {byts/ms/1e9:.3f} TB/s... ({byts/1e6:.1f} MB)", flush=True)

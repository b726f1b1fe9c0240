"""Full-size findings classifier (488 px, 16x16 grid, 2048 -> 512 -> 14): shape check and timing."""
import time, torch
from radialog_amd import synth
from radialog_amd.chexpert_model import ChexpertClassifier
from radialog_amd.engine import synth_getter
m = ChexpertClassifier(num_classes=14, dtype="f16", max_batch=8)
m.set_weight_getter(synth_getter(m.cfg, torch.device("cuda", 0), lora=False))
for B in (1, 1, 8, 1):
    x = synth.synth_images(B, 488, seed=3).cuda()
    out = m(x); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); out = m(x); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("B=%d logits %s finite=%s  per-call ms: %s" % (B, tuple(out.shape), bool(torch.isfinite(out).all()), ["%.2f" % t for t in ts]))

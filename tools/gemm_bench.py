"""Large-M GEMM throughput through rdx_gemm_test (includes weight packing; times the GEMM by differencing two iteration counts is
not possible through this entry, so this uses the prefill path instead): python tools/gemm_bench.py"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import time, torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
cfg = full_cfg()
B, T = 32, 160
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=256, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=True, seed=7).to(eng.device)
qf = synth.synth("u.qf", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=1, eos_id=-1, pad_id=0, use_graph=False); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): eng.generate(ids, qf, max_new=1, eos_id=-1, pad_id=0, use_graph=False)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
flops = 2 * 6.476e9 * B * T
print("prefill B=%d T=%d: %.1f ms -> %.3f PFLOP/s (%.1f %% of 2.5)" % (B, T, ms, flops / ms / 1e12, flops / ms / 1e12 / 2.5 * 100))
eng.close()

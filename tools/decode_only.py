#!/usr/bin/env python
"""One prefill + N greedy decode steps at batch B on the Vicuna-7B shapes (for rocprofv3 kernel traces of the decode step's kernel mix):
python tools/decode_only.py B N [xs16 = 1|0] [fp8 = 0|1] [dtype = bf16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radialog_amd import synth  # noqa: E402
from radialog_amd.config import full_cfg  # noqa: E402
from radialog_amd.engine import RdxEngine, synth_getter  # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
xs16 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fp8 = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
cfg = full_cfg()
T = 160
eng = RdxEngine(cfg, dtype=dtype, device=0, max_batch=B, max_len=(T + N + 64 + 31) // 32 * 32, lora=True, vision=False, weights_fp8=fp8)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
eng.set_option("xs16", xs16)
ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=(B > 1), seed=7).to(eng.device)
qf = synth.synth("t.qf_step", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
eng.generate(ids, qf, max_new=4, eos_id=-1, pad_id=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0)
torch.cuda.synchronize()
print(f"decode B={B} N={N} xs16={xs16} fp8={int(fp8)} {dtype}: {(time.perf_counter() - t0) * 1e3:.1f} ms incl. prefill")
eng.close()

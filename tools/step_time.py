#!/usr/bin/env python
"""Decode-step time per batch size (README's batch table, VERDICT r4 "next" 4): Vicuna-7B shapes, bf16 (or --dtype f16), prompt 160 tokens,
(a) the hipGraph step replayed at context ~168 (rdx_time unit 0), (b) the mean step of a 256-token greedy decode (contexts 160 .. 415, what
bench.py averages), with the HBM fraction (weights + KV of SURVEY 8d over 8 TB/s). `--ab` repeats every batch with the one-row-tile family
off (rdx_set_option xs16 0: the 32-row kernels of xstat32.hip). python tools/step_time.py [--batches 1,2,4,8,12,16,32] [--ab] [--fp8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from radialog_amd import synth  # noqa: E402
from radialog_amd.config import full_cfg  # noqa: E402
from radialog_amd.engine import RdxEngine, synth_getter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,3,4,8,12,16,32")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--ab", action="store_true")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--new", type=int, default=256)
    a = ap.parse_args()
    cfg = full_cfg()
    lc = cfg.llama
    T, N = 160, a.new
    wb = bench.llama_param_bytes(lc, lambda n, k: 1 if a.fp8 else 2)
    print(f"| batch | family | step at ctx {T + 8} (ms) | mean step over {N} tokens (ms) | HBM frac (weights + KV) | prefill ms |")
    print("|---|---|---|---|---|---|")
    for B in [int(x) for x in a.batches.split(",")]:
        eng = RdxEngine(cfg, dtype=a.dtype, device=0, max_batch=B, max_len=(T + N + 64 + 31) // 32 * 32, lora=True, vision=False, weights_fp8=a.fp8)
        eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
        ids = synth.synth_prompt_ids(B, T, vocab=lc.vocab, pad_rows=(B > 1), seed=7).to(eng.device)
        qf = synth.synth("t.qf_step", (B, 32, lc.qformer_dim), -1.0, 1.0).to(eng.device)
        for fam in ([1, 0] if (a.ab and 3 <= B <= 16) else [1]):
            eng.set_option("xs16", fam)
            eng.generate(ids, qf, max_new=8, eos_id=-1, pad_id=0)
            step = eng.time_unit(0, 20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.generate(ids, qf, max_new=1, eos_id=-1, pad_id=0)
            torch.cuda.synchronize()
            pre = (time.perf_counter() - t0) / 3
            eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                toks, _, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0)
            torch.cuda.synchronize()
            avg = ((time.perf_counter() - t0) / 2 - pre) / (N - 1) * 1e3
            kv = B * (T + N / 2.0) * 524288 + B * 524288
            frac = (wb + kv) / (avg * 1e-3) / 1e9 / bench.HBM_PEAK_GBS
            name = ("fp8 x fp8: " if a.fp8 else "") + ("xs16 (5 launches / layer)" if (fam and not a.fp8 and 3 <= B <= 16) else "chained (3 / layer)" if B <= 2 else
                    "row blocks (33-128 rows, 7 / layer)" if B > 32 else "xstat32 (7 launches / layer)")
            print(f"| {B} | {name} | {step:.3f} | {avg:.3f} | {frac * 100:.1f} % | {pre * 1e3:.1f} |", flush=True)
        eng.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""An EOS-ON workload (VERDICT r4 "next" 7): batch-32 greedy decode where rows really finish. The random-init model never emits token 2, so the
stop token is chosen from what it DOES emit: the run is decoded once with EOS off, then `eos_id` = the token at position `cut` of the row that
finishes last among the rows containing it ... simply: the most frequent token of the first half of the reports, which ends every row somewhere
inside the 256-token budget. Reported: the step at which the last row finished, the steps the engine executed before its poll noticed (poll
interval RDX_EOS_POLL, default 4; 16 = rounds 1-4) and the time of the generate call. python tools/eos_time.py [B = 32] [N = 256]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(B, N, eos):
    from radialog_amd import synth
    from radialog_amd.config import full_cfg
    from radialog_amd.engine import RdxEngine, synth_getter
    cfg = full_cfg()
    T = 160
    eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=B, max_len=(T + N + 64 + 31) // 32 * 32, lora=True, vision=False)
    eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=(B > 1), seed=7).to(eng.device)
    qf = synth.synth("t.qf_step", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0).to(eng.device)
    if eos is None:
        toks, _, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0)
        t = toks.cpu()
        vals, counts = t[:, : N // 2].flatten().unique(return_counts=True)
        # a token every row emits: candidates by frequency, first one present in all rows
        for v in vals[counts.argsort(descending=True)].tolist():
            if all((t[b] == v).any() for b in range(B)):
                first = [int((t[b] == v).nonzero()[0]) for b in range(B)]
                print(f"EOS {v} {max(first) + 1}")
                return
        print("EOS -1 0")
        return
    eng.generate(ids, qf, max_new=N, eos_id=eos, pad_id=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        toks, _, n = eng.generate(ids, qf, max_new=N, eos_id=eos, pad_id=0)
    torch.cuda.synchronize()
    print(f"RESULT steps_executed {n} ms {(time.perf_counter() - t0) / 3 * 1e3:.1f}")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    if len(sys.argv) > 3:
        run(B, N, None if sys.argv[3] == "find" else int(sys.argv[3]))
        sys.exit(0)
    out = subprocess.run([sys.executable, __file__, str(B), str(N), "find"], capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("EOS ")][-1].split()
    eos, last = int(line[1]), int(line[2])
    print(f"# EOS-on decode, batch {B}, budget {N} tokens, bf16: stop token {eos} (emitted by every row; the last row finishes at step {last})")
    print("| poll interval | steps executed | steps past the last EOS | generate() ms |")
    print("|---|---|---|---|")
    for poll in (16, 4, 1):
        env = dict(os.environ, RDX_EOS_POLL=str(poll))
        o = subprocess.run([sys.executable, __file__, str(B), str(N), str(eos)], capture_output=True, text=True, env=env).stdout
        r = [l for l in o.splitlines() if l.startswith("RESULT")][-1].split()
        print(f"| {poll}{' (rounds 1-4)' if poll == 16 else ' (round 5 default)' if poll == 4 else ''} | {r[2]} | {int(r[2]) - last} | {r[4]} |")

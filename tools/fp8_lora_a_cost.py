#!/usr/bin/env python
"""What quantising the LoRA-A rows costs in the fp8 configuration (VERDICT r4 "next" 5c): the CPU oracle's fp8 restatement with the LoRA-A matrices in e4m3
(what the engine does: they ride the QKV weight as 16 extra rows) against the same oracle with LoRA-A kept in the model dtype, and both against the
un-quantised oracle -- production width, `layers` decoder layers, one 96-token prompt, 8 greedy steps, bf16. CPU only.  python tools/fp8_lora_a_cost.py [layers=4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cpu  # noqa: E402
from radialog_amd import synth  # noqa: E402
from radialog_amd.config import LlamaCfg  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.set_num_threads(min(32, os.cpu_count() or 8))
lc = LlamaCfg(layers=layers, qformer_dim=192)
W = synth.make_weights(synth.llama_specs(lc, lora=True))
ids = synth.synth_prompt_ids(1, 96, vocab=lc.vocab, img_offset=6, seed=5)
qf = synth.synth("t.qf_loraA", (1, 32, lc.qformer_dim), -1.0, 1.0)
N = 8
with torch.no_grad():
    base = ref_cpu.LlamaOracle(W, lc, torch.bfloat16, lora=True).generate_greedy(ids, qf, max_new=N, eos_id=-1)
    q = ref_cpu.LlamaOracle(W, lc, torch.bfloat16, lora=True, fp8=True)
    q.force_a8 = True
    full = q.generate_greedy(ids, qf, max_new=N, eos_id=-1)
    for k in [k for k in q.W8 if k.endswith("lora_A.weight")]:
        del q.W8[k]                                                    # LoRA-A back in the model dtype (and its input un-quantised)
    keep = q.generate_greedy(ids, qf, max_new=N, eos_id=-1)


def err(a, b):
    n = 0
    for s in range(N):
        if s and int(a["tokens"][0, s - 1]) != int(b["tokens"][0, s - 1]):
            break
        n = s + 1
    return max(float((a["scores"][s].float() - b["scores"][s].float()).abs().max()) for s in range(n)), n


e_full, n1 = err(full, base)
e_keep, n2 = err(keep, base)
e_delta, n3 = err(full, keep)
print(f"production width, {layers} layers, bf16, {N} greedy steps (compared while the token paths agree):")
print(f"  fp8 oracle (LoRA-A e4m3, the engine's scheme) vs un-quantised: worst logit error {e_full:.3f} over {n1} steps")
print(f"  fp8 oracle with LoRA-A in the model dtype       vs un-quantised: worst logit error {e_keep:.3f} over {n2} steps")
print(f"  the two fp8 variants against each other:                          worst logit error {e_delta:.3f} over {n3} steps")

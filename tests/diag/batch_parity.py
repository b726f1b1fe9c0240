import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sys, torch
from radialog_amd import synth
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine, synth_getter
from oracle import ref_cpu
cfg = small_cfg()
specs = {}
specs.update(synth.qformer_specs(cfg.qformer)); specs.update(synth.llama_specs(cfg.llama, lora=True)); specs.update(synth.vision_specs(cfg.vision))
W = synth.make_weights(specs)
dtype = "f16"; dt = torch.float16
for B in [int(x) for x in sys.argv[1:]]:
    T, N = 96, 6
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=6, pad_rows=True, seed=3)
    qf = synth.synth("t.qf18", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    eng = RdxEngine(cfg, dtype=dtype, device=0, max_batch=B, max_len=128, lora=True, vision=False)
    eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
    with torch.no_grad():
        ref = ref_cpu.LlamaOracle(W, cfg.llama, dt, lora=True).generate_greedy(ids, qf, max_new=N, eos_id=-1, pad_id=0)
    toks, scores, n = eng.generate(ids, qf, max_new=N, eos_id=-1, pad_id=0, output_scores=True, use_graph=False)
    toks = toks.cpu().long()
    bad = []
    for b in range(B):
        for s in range(N):
            err = float((scores[s, b].float().cpu() - ref["scores"][s][b].float()).abs().max())
            if toks[b, s] != ref["tokens"][b, s] or err > 0.01:
                bad.append((b, s, round(err, 4), float(ref["margins"][s, b])))
                break
    print("B", B, "bad rows (row, step, logit err, margin):", bad[:8])
    eng.close()

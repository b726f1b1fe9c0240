"""Beam search at the 7B shape: python tools/beam_time.py [groups] [beams] [max_new]  -- time per forward with the
suffix-only _reorder_cache against the move-everything A/B leg (RDX_BEAM_FULLCOPY=1)."""
import os, sys, time, torch
from radialog_amd import synth
from radialog_amd.config import full_cfg
from radialog_amd.engine import RdxEngine, synth_getter
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = 160
cfg = full_cfg()
eng = RdxEngine(cfg, dtype="bf16", device=0, max_batch=G * K, max_len=(T + N + 63) // 32 * 32, lora=True, vision=False)
eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
ids = synth.synth_prompt_ids(G, T, vocab=cfg.llama.vocab, pad_rows=True, seed=7)
qf = synth.synth("u.qf", (G, 32, cfg.llama.qformer_dim), -1.0, 1.0)
res = {}
for full in ("1", "0"):
    os.environ["RDX_BEAM_FULLCOPY"] = full
    eng.beam_search(ids, qf, K, 8, eos_id=-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, lens, sc, _, n = eng.beam_search(ids, qf, K, N, eos_id=-1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[full] = (toks.clone(), sc.clone())
    print("beam search %d prompts x %d beams, %d new tokens, %s: %.1f ms total, %.3f ms per forward" %
          (G, K, N, "every generated position moved" if full == "1" else "diverging suffix only", dt * 1e3, dt * 1e3 / n))
print("identical results:", torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1]))
eng.close()

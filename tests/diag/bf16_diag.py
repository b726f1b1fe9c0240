"""Diagnostic (not a test): where does the HIP path's distance from the exactly-accumulated evaluation come from?
Production width, 1 layer: layer-0 K/V cache, last-position hidden state and logits of the prefill, then decode steps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu
from radialog_amd import synth
from radialog_amd.config import LlamaCfg, RaDialogCfg
from radialog_amd.engine import RdxEngine, synth_getter
import ctypes as C

DT = {"f16": torch.float16, "bf16": torch.bfloat16}
cfg = RaDialogCfg(llama=LlamaCfg(layers=1, qformer_dim=192))
W = synth.make_weights(synth.llama_specs(cfg.llama, lora=True))
T, N = 96, 3


def stats(a, b):
    d = (a.float() - b.float()).abs()
    return f"max {float(d.max()):.4g} mean {float(d.mean()):.3g}"


for dtype in sys.argv[1:] or ["bf16", "f16"]:
    for B in (1, 8):
        ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=6, pad_rows=False, seed=5)
        qf = synth.synth("t.qf20", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
        km = ids.ne(0).long()
        res = {}
        for name, exact in (("oracle", False), ("exact", True)):
            o = ref_cpu.LlamaOracle(W, cfg.llama, DT[dtype], lora=True, exact=exact)
            with torch.no_grad():
                lg, past, _ = o.forward(o.embed(ids, qf), km, ref_cpu.positions_from_mask(km))
                g = o.generate_greedy(ids, qf, max_new=N, eos_id=-1)
            res[name] = dict(logits=lg[:, -1], k=past[0][0], v=past[0][1], hid=o.last_hidden[:, -1] if False else None, gen=g)
            o2 = ref_cpu.LlamaOracle(W, cfg.llama, DT[dtype], lora=True, exact=exact)
            with torch.no_grad():
                o2.forward(o2.embed(ids, qf), km, ref_cpu.positions_from_mask(km))
            res[name]["hid"] = o2.last_hidden[:, -1]
        eng = RdxEngine(cfg, dtype=dtype, device=0, max_batch=B, max_len=128, lora=True, vision=False)
        eng.load_weights(synth_getter(cfg, eng.device, lora=True), vision=False)
        toks, lg = eng.prefill(ids, qf, max_new=4)
        hid = torch.empty(B, cfg.llama.hidden, dtype=DT[dtype], device=eng.device)
        eng.lib.rdx_hidden_read(eng.ctx, C.c_void_p(hid.data_ptr())); eng.sync()
        k0 = eng.kv_read(0, 0, B)[:, :, :T].cpu()
        v0 = eng.kv_read(0, 1, B)[:, :, :T].cpu()
        print(f"== {dtype} B={B}  prefill")
        for nm in ("oracle", "exact"):
            r = res[nm]
            print(f"   vs {nm:6s}: K {stats(k0, r['k'])} | V {stats(v0, r['v'])} | hidden {stats(hid.cpu(), r['hid'])} | logits {stats(lg.cpu(), r['logits'])}")
        print(f"   oracle vs exact: K {stats(res['oracle']['k'], res['exact']['k'])} | V {stats(res['oracle']['v'], res['exact']['v'])} | "
              f"hidden {stats(res['oracle']['hid'], res['exact']['hid'])} | logits {stats(res['oracle']['logits'], res['exact']['logits'])}")
        t, sc, n = eng.generate(ids, qf, max_new=N, eos_id=-1, output_scores=True)
        for s in range(N):
            print(f"   step {s}: hip-oracle {stats(sc[s].cpu(), res['oracle']['gen']['scores'][s])} | hip-exact {stats(sc[s].cpu(), res['exact']['gen']['scores'][s])} | "
                  f"oracle-exact {stats(res['oracle']['gen']['scores'][s], res['exact']['gen']['scores'][s])}")
        eng.close()

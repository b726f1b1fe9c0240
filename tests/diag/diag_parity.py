"""Print actual parity errors (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from radialog_amd import synth
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine, synth_getter
from oracle import ref_cpu

cfg = small_cfg()
specs = {}
specs.update(synth.vision_specs(cfg.vision)); specs.update(synth.qformer_specs(cfg.qformer)); specs.update(synth.llama_specs(cfg.llama))
cpu_w = synth.make_weights(specs)
for dtn, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
    eng = RdxEngine(cfg, dtype=dtn, max_batch=4, max_len=256)
    eng.load_weights(synth_getter(cfg, eng.device))
    img = synth.synth_images(2, cfg.vision.img)
    rq, remb = ref_cpu.forward_image(img, cpu_w, cfg)
    q, emb = eng.encode_image(img.cuda())
    rl = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(dtn, "enc rel_l2 emb", rl(emb.cpu(), remb), "qformer", rl(q.cpu(), rq), "max|q|", float(rq.abs().max()))
    B, T, N = 3, 72, 32
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, img_offset=6, seed=33)
    ids[1] = torch.cat([torch.zeros(5, dtype=torch.long), ids[1, : T - 5]])
    ids[2][ids[2] == 32000] = 99
    qf = synth.synth("t.qf2", (B, 32, cfg.llama.qformer_dim), -1.0, 1.0)
    orc = ref_cpu.LlamaOracle(cpu_w, cfg.llama, dt)
    with torch.no_grad():
        ref = orc.generate_greedy(ids, qf, max_new=N, eos_id=-1)
    for g in (False, True):
        toks, scores, n = eng.generate(ids, qf, max_new=N, eos_id=-1, output_scores=True, use_graph=g)
        toks = toks.cpu().long()
        match = (toks == ref["tokens"]).float().mean().item()
        errs = [float((scores[s].float().cpu() - ref["scores"][s].float()).abs().max()) for s in range(N)]
        print(dtn, "graph" if g else "eager", "token match frac", match, "max logit err", max(errs), "min margin", float(ref["margins"].min()),
              "logit absmax", float(ref["scores"][0].float().abs().max()))
        print("   tokens gpu", toks[0, :10].tolist(), "ref", ref["tokens"][0, :10].tolist())
    eng.close()

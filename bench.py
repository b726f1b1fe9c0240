#!/usr/bin/env python
"""bench.py -- RaDialog hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md "Measurement").

A "step" is one pass of the hot path over one batch of synthetic input on each rank:
    BioViL-T encode (448x448 synthetic CXR) -> Q-Former -> img_proj + <IMG> splice -> Vicuna-7B prefill (160-token
    prompt with 32 <IMG> slots) -> 256 greedy tokens (eos disabled so every report is exactly 256 tokens).
Default workload = BASELINE.json configs[1]: 1 x MI355X, batch 1, bf16, random-init weights of the real architecture.
`--batch 32` gives configs[2] (batch-32 decode, hipGraph step). With N > 1 ranks (one process per GPU, launched by
torch.distributed.run) every rank runs the same per-GPU batch on its own shard of images (weak scaling) and the
generated token ids are all-gathered over RCCL/xGMI at the end of each step -- the only collective on the path.

Rank 0 prints ONE JSON line. `value` = reports/s of the whole job with inputs resident in HBM. Added objects:
  roofline     the dominant kernel = the gate/up SwiGLU weight-streaming GEMV (45 % of decode bytes): algorithmic bytes per
               launch / its average launch duration measured with HIP events on the library's stream (rdx_time).
  cpu_baseline the CPU oracle (oracle/ref_cpu.py, kind "port") timed on the host cores on a bounded sample of the same
               workload, extrapolated to reports/s.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="reports per GPU per step (1 = configs[1], 32 = configs[2])")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=160)
    ap.add_argument("--fp8", action="store_true", help="decoder GEMM weights in e4m3 + per-row scale (BASELINE configs[4] weight path); "
                    "not the parity configuration: its oracle is the reference math on the fake-quantised weights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    return ap.parse_args()


def pmc_traffic(batch, dtype, fp8=False):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3
    passes, gfx950 read-side doubling): measured offline because counters cannot be read from inside the timed process;
    the summary lives in profiles/r01_pmc_hbm_traffic.md and its machine-readable twin profiles/r01_pmc.json.
    Only valid for the configurations it was measured on (bf16: batch 1 with or without fp8 weights, batch 32) -> null otherwise."""
    if batch not in (1, 32) or dtype != "bf16" or (batch == 32 and fp8):
        return None
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc.json")) as f:
            d = json.load(f)
        for k, v in d.items():
            if ("W8" in k) == bool(fp8) and (f"B={batch}," in k):
                return v["traffic_bytes"]
        return None
    except Exception:
        return None


def _pick_threads():
    """torch's intra-op pool at os.cpu_count() threads can be far slower than a smaller pool on many-core hosts
    (sync overhead on decode-sized ops); calibrate on a decode-shaped matmul and use the fastest setting."""
    import os as _os
    ncpu = _os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, 128, ncpu) if c <= ncpu})
    w = torch.randn(4096, 4096).to(torch.bfloat16)
    x = torch.randn(1, 4096).to(torch.bfloat16)
    best, best_t = cands[0], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        torch.nn.functional.linear(x, w)
        t0 = time.time()
        for _ in range(3):
            torch.nn.functional.linear(x, w)
        t = time.time() - t0
        if t < best_t:
            best, best_t = c, t
    torch.set_num_threads(best)
    return best


def cpu_baseline(cfg, dtype_name, prompt_len, new_tokens):
    """Time the CPU oracle on a BOUNDED sample of the same workload: the full-size encode of ONE image, and 1 of the 32
    decoder layers (+ final norm + lm_head) for the 160-token prefill and 3 decode steps; extrapolate linearly in
    layers and tokens. Target: 10-30 s of CPU work."""
    from oracle import ref_cpu
    from radialog_amd import synth
    from radialog_amd.config import LlamaCfg
    threads = _pick_threads()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype_name]
    t_all = time.time()
    sample_layers, sample_steps = 1, 3
    lc = cfg.llama
    sub = LlamaCfg(vocab=lc.vocab, hidden=lc.hidden, inter=lc.inter, layers=sample_layers, heads=lc.heads)
    with torch.no_grad():
        Wv = synth.make_weights({**synth.vision_specs(cfg.vision), **synth.qformer_specs(cfg.qformer)})
        img = synth.synth_images(1, cfg.vision.img)
        t0 = time.time(); q, _ = ref_cpu.forward_image(img, Wv, cfg); t_enc = time.time() - t0
        del Wv
        Wl = synth.make_weights(synth.llama_specs(sub, lora=True), dtype=dt)
        orc = ref_cpu.LlamaOracle(Wl, sub, dt, lora=True)
        del Wl
        ids = synth.synth_prompt_ids(1, prompt_len, vocab=lc.vocab)
        km = ids.ne(0).long()
        t0 = time.time()
        x = orc.embed(ids, q)
        logits, past, _ = orc.forward(x, km, ref_cpu.positions_from_mask(km))
        t_prefill = time.time() - t0
        E = orc.W["model.embed_tokens.weight"]
        t_steps = []
        for s in range(sample_steps):
            t0 = time.time()
            nxt = logits[:, -1].argmax(-1)
            km = torch.cat([km, km.new_ones(1, 1)], -1)
            xx = torch.nn.functional.embedding(nxt[:, None], E)
            logits, past, _ = orc.forward(xx, km, ref_cpu.positions_from_mask(km)[:, -1:], past)
            t_steps.append(time.time() - t0)
        t_step = min(t_steps)
        # lm_head + final norm alone (not proportional to the layer count)
        h = torch.randn(1, 1, lc.hidden).to(dt)
        torch.nn.functional.linear(h, orc.W["lm_head.weight"])
        t0 = time.time()
        torch.nn.functional.linear(ref_cpu.rmsnorm(h, orc.W["model.norm.weight"], lc.rms_eps), orc.W["lm_head.weight"])
        t_head = time.time() - t0
    scale = lc.layers / sample_layers
    t_prefill_full = max(t_prefill - t_head, 0.0) * scale + t_head
    t_step_full = max(t_step - t_head, 0.0) * scale + t_head
    t_report = t_enc + t_prefill_full + new_tokens * t_step_full
    return {
        "value": 1.0 / t_report, "unit": "reports/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
        "sample": (f"oracle/ref_cpu.py on {threads} torch threads (fastest of a calibration sweep; host has {os.cpu_count()} cpus): "
                   f"1 image full-size encode ({t_enc:.2f} s, cold) + {sample_layers}/{lc.layers} decoder layers "
                   f"({dtype_name}) for prefill T={prompt_len} ({t_prefill:.2f} s) and {sample_steps} decode steps "
                   f"(best {t_step*1e3:.0f} ms), lm_head {t_head*1e3:.0f} ms; extrapolated x{scale:.0f} layers, "
                   f"{new_tokens} tokens -> {t_report:.1f} s/report; sample wall {time.time()-t_all:.0f} s"),
        "s_per_token": t_step_full, "s_encode": t_enc, "s_prefill": t_prefill_full,
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run (also at --nproc-per-node 1)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from radialog_amd import synth
    from radialog_amd.config import full_cfg
    from radialog_amd.engine import RdxEngine, synth_getter
    from radialog_amd.shard import allgather_tokens

    cfg = full_cfg()
    B, T, N = args.batch, args.prompt_len, args.new_tokens
    max_len = (T + N + 64 + 31) // 32 * 32
    eng = RdxEngine(cfg, dtype=args.dtype, device=local_rank, max_batch=B, max_len=max_len, lora=True, weights_fp8=args.fp8)
    eng.load_weights(synth_getter(cfg, eng.device, lora=True))

    # this rank's shard of the (synthetic) image batch and prompts, resident in HBM before the timed region
    img = synth.synth_images(B, cfg.vision.img, seed=16 + rank).to(eng.device)
    ids = synth.synth_prompt_ids(B, T, vocab=cfg.llama.vocab, pad_rows=(B > 1), seed=7 + rank).to(eng.device)
    use_graph = not args.no_graph

    out_q = None

    def step():
        nonlocal out_q
        q, _ = eng.encode_image(img, want_image_embeds=False)
        out_q = q
        toks, _, n = eng.generate(ids, q, max_new=N, eos_id=-1, pad_id=0, use_graph=use_graph)
        if world > 1:
            return allgather_tokens(toks, world)
        return toks

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    assert out.shape[0] == B * world and out.shape[1] == N

    if rank == 0:
        # encoder ms/img (second half of the metric), measured alone
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.encode_image(img, want_image_embeds=False)
        enc_ms = (time.perf_counter() - t1) / 5 / B * 1e3
        # dominant kernel: gate/up SwiGLU GEMV, HIP events on the library's stream
        lc = cfg.llama
        gu_ms = eng.time_unit(1, 10)
        def wbytes(n_rows, k):        # bytes per streamed weight element of a decode projection (mirrors the library's dispatch)
            if not args.fp8:
                return 2
            tiles = (n_rows + 15) // 16
            lds = B <= 16 and B * k * 2 <= 32 * 1024                     # batch-1..4 GEMV with LDS-staged activations
            s32 = 16 < B <= 32 and tiles > 512 and k % 512 == 0          # skinny32.hip
            xs = 3 <= B <= 32 and ((tiles >= 512 and k == 4096) or (128 <= tiles <= 512 and k in (4096, 11008)))   # xstat32.hip
            return 1 if (lds or s32 or xs) else 2
        wb = wbytes(2 * lc.inter, lc.hidden)
        gu_bytes = 2 * lc.inter * lc.hidden * wb + B * lc.hidden * 2 + lc.hidden * 2 + B * lc.inter * 2
        # whole-decode average: one report minus its encode and its prefill (+ first token), over the N-1 graph-replayed steps
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(3):
            eng.generate(ids, out_q, max_new=1, eos_id=-1, pad_id=0, use_graph=use_graph)
        torch.cuda.synchronize()
        prefill_ms = (time.perf_counter() - t2) / 3 * 1e3
        avg_step_ms = (elapsed / args.steps * 1e3 - enc_ms * B - prefill_ms) / max(N - 1, 1)
        kv_bytes = B * (T + N / 2.0) * 524288 + B * 524288          # 2 x 32 layers x 4096 x 2 B per cached token
        step_ms = eng.time_unit(0, 20)
        L_avg = T + 64      # rdx_time(0) replays from the state left by the last generate (slot ~ T+N) -- report as measured
        H_, I_ = lc.hidden, lc.inter
        step_bytes = (32 * (3 * H_ * H_ * wbytes(3 * H_ + 16, H_) + H_ * H_ * wbytes(H_, H_) + 2 * I_ * H_ * wbytes(2 * I_, H_)
                            + H_ * I_ * wbytes(H_, I_)) + lc.vocab * H_ * wbytes(lc.vocab, H_) + (32 * 2 * H_ + H_) * 2)
        roof = {
            "bound": "hbm", "kernel": (f"xstat32_k<{args.dtype},EPI_SILU_MUL{',W8' if wb == 1 else ''}> (gate/up SwiGLU, activation-stationary batch 3-32 GEMM)" if B > 2 else
                                       f"skinny_gemm_k<{args.dtype},MT,EPI_SILU_MUL,NORM{',W8' if wb == 1 else ''}> (gate/up SwiGLU GEMV)"),
            "achieved": gu_bytes / (gu_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gu_bytes / (gu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(B, args.dtype, args.fp8),
            "bytes_per_launch": gu_bytes, "us_per_launch": gu_ms * 1e3,
            "decode_step_ms": step_ms, "decode_step_weight_GBs": step_bytes / (step_ms * 1e-3) / 1e9,
            "decode_step_frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # average over the timed reports: (weights + KV read/write per SURVEY 8(d)) / mean step time / 8 TB/s
            "prefill_ms": prefill_ms, "decode_avg_step_ms": avg_step_ms,
            "decode_avg_frac": (step_bytes + kv_bytes) / (avg_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        res = {
            "metric": "reports_per_sec (448px CXR encode + 160-tok prefill + 256-tok greedy decode)",
            "value": args.steps * B * world / elapsed, "unit": "reports/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype + ("+fp8w" if args.fp8 else ""), "data": "synthetic",
            "config": {"workload": f"configs[{1 if B == 1 else 2}]: per-GPU batch {B} BioViL-T(ResNet-50)+Q-Former encode 448px, "
                                   f"Vicuna-7B prefill T={T}, {N}-token greedy decode (LoRA r=8 un-merged, hipGraph step={use_graph}"
                                   + (", fp8 e4m3 decoder weights + per-row scale [configs[4] weight path]" if args.fp8 else "") + ")",
                       "per_gpu_batch": B, "global_batch": B * world, "prompt_len": T, "new_tokens": N,
                       "parallelism": f"dp{world}", "weights": "random-init (deterministic generator)"},
            "encoder_ms_per_img": enc_ms,
            "tokens_per_s": args.steps * B * world * N / elapsed,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, args.dtype, T, N)
        print(json.dumps(res), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

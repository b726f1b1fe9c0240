#!/usr/bin/env python
"""profiles/rNN_pmc.json + rNN_pmc_hbm_traffic.md from the six text summaries `tools/profile_r05.sh pmc` (r03 / r04: profile_r03.sh) leaves in gpurun_out/prof/
(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, one counter per pass, of bench.py at batch 1 / 32 / 32 fp8):
python tools/pmc_to_json.py r05. traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH doubled per the gfx950 correction of
MI355X_MICROARCH.md; `pick` = which statistic of a kernel's dispatches stands for the typical launch. Round 5: the file is stamped with the hash of
the kernel sources it was taken on (`_tree` = radialog_amd.build.source_hash(); bench.py flags a replayed figure whose tree is not the running one) and
every entry carries `algorithmic_bytes` AT THE PROFILED STATE (the batch-32 passes run a 280-token prompt + 8 tokens, so the decode attention is
profiled at the mean context 288 bench.py prices it at; flash_prefill_k / the prefill GEMMs are then those of a 280-token prompt)."""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof")
# key -> (run, kernel-name regex, statistic, note)
KERNELS = {
    "decode_chain_k B=1 bf16": ("b1", r"decode_chain_kIDF16bLb0", "max", "typical launch = down_proj(l) + RMSNorm + QKV(l+1) (max over the dispatches; the average includes the short last-layer launch); algorithmic 191.04 MB"),
    "attn_oproj16_k B=1 bf16": ("b1", r"attn_oproj16_kIDF16bLb0", "avg", "fused decode attention + o_proj at context ~165: o_proj weights 33.55 MB + KV 2 x 165 x 8 KiB + new-token operands"),
    "gate_up B=1 bf16": ("b1", r"skinny_gemm_k<bool _Accum, int, E, 4, true, 4, true, false>", "avg", "skinny_gemm_k<EPI_SILU_MUL>; algorithmic 180.4 MB"),
    "gate_up B=32 bf16": ("b32", r"xstat32_kIDF16bLi4ELb0ELb0", "avg", "xstat32_k<EPI_SILU_MUL>; algorithmic 181.3 MB"),
    "qkv B=32 bf16": ("b32", r"xstat32_kIDF16bLi0ELb0ELb0", "avg", "xstat32_k<EPI_NONE> QKV + LoRA-A rows; algorithmic 100.8 MB"),
    "down B=32 bf16": ("b32", r"xsplit32_kIDF16bLi344", "avg", "xsplit32_k down_proj, K split 4 ways; algorithmic 90.2 MB + activations"),
    "o_proj B=32 bf16": ("b32", r"xsplit32_kIDF16bLi128", "avg", "xsplit32_k o_proj, K split 2 ways; algorithmic 33.55 MB"),
    "decode_attention_k B=32 bf16": ("b32", r"decode_attention_kIDF16bLi4", "avg", "batch-32 decode attention at contexts 280 .. 295 (a 280-token prompt + 8 tokens: the mean context of the benchmark's decode is 288): KV 32 x 32 x 2 x 288 x 256 B = 151 MB algorithmic"),
    "flash_prefill_k B=32 bf16": ("b32", r"flash_prefill_kIDF16b", "avg", "K twice + V once per 64-query block, five blocks per (row, head) at T = 280; algorithmic = Q, K, V read + O written"),
    "gemm_dma256_k gate_up prefill B=32 bf16": ("b32", r"gemm_dma256_kIDF16bLi4ELi8", "avg", "8960 x 22016 x 4096 bf16 (32 prompts of 280 tokens), XCD-compact tile order"),
    "gemm8_256_k gate_up prefill B=32 fp8": ("b32fp8", r"gemm8_256_kIDF16bLi4ELi8", "avg", "8960 x 22016 x 4096 e4m3 x e4m3 (32 prompts of 280 tokens), XCD-compact tile order"),
    "gate_up B=32 bf16 fp8": ("b32fp8", r"xstat32_kIDF16bLi4ELb1ELb1", "avg", "xstat32_k<W8, A8>; algorithmic 90.7 MB"),
    "decode_attention_k B=32 bf16 fp8": ("b32fp8", r"decode_attention_kIDF16bLi4", "avg", "as the bf16 run (the KV cache stays bf16)"),
}


ALGO = {   # algorithmic bytes per launch at the profiled state (SURVEY 8d shapes; bf16 = 2 B, e4m3 = 1 B)
    "decode_chain_k B=1 bf16": 191043104, "attn_oproj16_k B=1 bf16": 33554432 + 2 * 192 * 8192 + 5 * 8192, "gate_up B=1 bf16": 180363264 + 8192 + 22016,
    "gate_up B=32 bf16": 180363264 + 262144 + 704512, "qkv B=32 bf16": 100794368 + 262144 + 787456, "down B=32 bf16": 90177536 + 704512 + 2097152,
    "o_proj B=32 bf16": 33554432 + 262144 + 1048576, "decode_attention_k B=32 bf16": 32 * (2 * 288 * 4096 * 2 + 6 * 4096 * 2),
    "flash_prefill_k B=32 bf16": 4 * 32 * 280 * 4096 * 2, "gemm_dma256_k gate_up prefill B=32 bf16": 8960 * 4096 * 2 + 22016 * 4096 * 2 + 8960 * 11008 * 2,
    "gemm8_256_k gate_up prefill B=32 fp8": 8960 * 4096 + 22016 * 4096 + 8960 * 11008 * 2, "gate_up B=32 bf16 fp8": 90181632 + 131072 + 704512,
    "decode_attention_k B=32 bf16 fp8": 32 * (2 * 288 * 4096 * 2 + 6 * 4096 * 2),
}


def table(run, ctr):
    rows = {}
    path = os.path.join(SRC, f"pmc_{run}_{ctr}.txt")
    for line in open(path):
        m = re.match(rf"{ctr}\s+n=\s*(\d+) avg=\s*([\d.]+) min=\s*([\d.]+) max=\s*([\d.]+)\s+(.*)", line)
        if m:
            rows[m.group(5).strip()] = {"n": int(m.group(1)), "avg": float(m.group(2)), "min": float(m.group(3)), "max": float(m.group(4))}
    return rows, open(path).read()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    out, md = {}, [f"# HBM / fabric traffic per kernel launch, round {tag[1:]} tree: rocprofv3 PMC passes of bench.py (tools/profile_r03.sh pmc, one counter per pass, --kernel-trace only)\n",
                   "FETCH_SIZE / WRITE_SIZE are KiB per dispatch as rocprofv3 reports them; traffic = 2 x FETCH + WRITE (gfx950 correction for FETCH). Machine-readable twin: "
                   f"`profiles/{tag}_pmc.json` (read by bench.py for `roofline.traffic`).\n"]
    cache = {}
    for run in ("b1", "b32", "b32fp8"):
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cache[(run, ctr)], text = table(run, ctr)
            md.append(f"## bench.py {run} -- {ctr}\n```\n" + "\n".join(l[:170] for l in text.splitlines()[1:26]) + "\n```\n")
    for key, (run, pat, stat, note) in KERNELS.items():
        f = next((v for k, v in cache[(run, "FETCH_SIZE")].items() if re.search(pat, k)), None)
        w = next((v for k, v in cache[(run, "WRITE_SIZE")].items() if re.search(pat, k)), None)
        if f is None or w is None:
            continue
        out[key] = {"FETCH_SIZE_KiB": f[stat], "WRITE_SIZE_KiB": w[stat], "traffic_bytes": int((2 * f[stat] + w[stat]) * 1024), "dispatches": f["n"], "note": note,
                    "algorithmic_bytes": ALGO.get(key)}
    sys.path.insert(0, REPO)
    from radialog_amd import build as _b
    try:                                     # the hash the GPU box computed next to the passes; else this tree's (run it on the tree the passes were taken on)
        out["_tree"] = open(os.path.join(SRC, "pmc_tree.txt")).read().strip() or _b.source_hash()
    except OSError:
        out["_tree"] = _b.source_hash()
    json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    open(os.path.join(REPO, "profiles", f"{tag}_pmc_hbm_traffic.md"), "w").write("\n".join(md))
    for k, v in out.items():
        if isinstance(v, dict):
            print(f"{k:46s} {v['traffic_bytes'] / 1e6:9.1f} MB" + (f"  ({v['traffic_bytes'] / v['algorithmic_bytes']:.3f} x algorithmic)" if v.get("algorithmic_bytes") else ""))


if __name__ == "__main__":
    main()

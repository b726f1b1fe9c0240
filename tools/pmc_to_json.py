#!/usr/bin/env python
"""profiles/rNN_pmc.json + rNN_pmc_hbm_traffic.md from the six text summaries `tools/profile_r03.sh pmc` leaves in gpurun_out/prof/
(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, one counter per pass, of bench.py at batch 1 / 32 / 32 fp8):
python tools/pmc_to_json.py r04. traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH doubled per the gfx950 correction of
MI355X_MICROARCH.md; `pick` = which statistic of a kernel's dispatches stands for the typical launch."""
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
    "decode_attention_k B=32 bf16": ("b32", r"decode_attention_kIDF16bLi4", "avg", "batch-32 decode attention at context ~165 (8 new tokens behind a 160-token prompt): KV 32 x 32 x 2 x 165 x 256 B = 86.5 MB algorithmic"),
    "flash_prefill_k B=32 bf16": ("b32", r"flash_prefill_kIDF16b", "avg", "K twice + V once per 64-query block, three blocks per (row, head); algorithmic 167.8 MB"),
    "gemm_dma256_k gate_up prefill B=32 bf16": ("b32", r"gemm_dma256_kIDF16bLi4ELi8", "avg", "5120 x 22016 x 4096 bf16, XCD-compact tile order; algorithmic 222 MB in + 113 MB out"),
    "gemm8_256_k gate_up prefill B=32 fp8": ("b32fp8", r"gemm8_256_kIDF16bLi4ELi8", "avg", "5120 x 22016 x 4096 e4m3 x e4m3, XCD-compact tile order; algorithmic 111 MB in + 113 MB out"),
    "gate_up B=32 bf16 fp8": ("b32fp8", r"xstat32_kIDF16bLi4ELb1ELb1", "avg", "xstat32_k<W8, A8>; algorithmic 90.7 MB"),
    "decode_attention_k B=32 bf16 fp8": ("b32fp8", r"decode_attention_kIDF16bLi4", "avg", "as the bf16 run (the KV cache stays bf16)"),
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
        out[key] = {"FETCH_SIZE_KiB": f[stat], "WRITE_SIZE_KiB": w[stat], "traffic_bytes": int((2 * f[stat] + w[stat]) * 1024), "dispatches": f["n"], "note": note}
    json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    open(os.path.join(REPO, "profiles", f"{tag}_pmc_hbm_traffic.md"), "w").write("\n".join(md))
    for k, v in out.items():
        print(f"{k:46s} {v['traffic_bytes'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()

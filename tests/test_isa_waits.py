"""Build-quality guard (CPU: hipcc cross-compiles gfx950 without a GPU): the software-pipelined LDS-DMA GEMM loops must contain only the hand-placed counted
waits. hipcc (SIInsertWaitcnts) puts `s_waitcnt vmcnt(0)` in front of an LDS read whose IR load lost its TBAA tag while LDS-DMA is in flight -- the ring then
collapses to one stage and nothing fails functionally (gemm8_k ran a whole round that way: a single fp8 prompt's prefill 10.4 ms instead of 7.6 ms,
profiles/r04_fp8_prefill_mx.md). The test compiles the two GEMM units to assembly and scans every loop that holds both `global_load_lds` and MFMAs."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "radialog_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _kernels(asm):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s*(;.*)?$", l)]
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        yield lines[a].split(":")[0], lines[a:b]


def _dma_loops(body):
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loop = body[labels[m.group(1)]:i]
            if any("global_load_lds" in x for x in loop) and any("v_mfma" in x for x in loop):
                yield loop


def _drains_in_front_of_lds_reads(loop):
    """`s_waitcnt vmcnt(0)` outside inline asm with a ds_read among the next three instructions; the K-group rescale path of the fp8 kernels (a rare
    branch that reads scales from global memory and legitimately waits for them) is recognised by the global_load in front of the wait."""
    bad = 0
    for k, x in enumerate(loop):
        if "s_waitcnt vmcnt(0)" not in x:
            continue
        nxt = [y for y in loop[k + 1:k + 8] if y.strip() and not y.strip().startswith(";")][:3]
        prev = [y for y in loop[max(0, k - 40):k]]
        if any("ds_read" in y for y in nxt) and not any(re.search(r"global_load_dword\b|global_load_dwordx[24]\b(?!.*lds)", y) and "global_load_lds" not in y for y in prev):
            bad += 1
    return bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit,families", [("gemm8.hip", ("gemm8_kI", "gemm8_256_kI")), ("gemm_dma.hip", ("gemm_dma_kI", "gemm_dma256_kI"))])
def test_lds_dma_loops_hold_no_compiler_drain(unit, families, tmp_path):
    out = tmp_path / (unit + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I" + CSRC, os.path.join(CSRC, unit), "-o", str(out)]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode == 0 and out.exists(), p.stderr[-2000:]
    seen = {f: 0 for f in families}
    offenders = []
    for name, body in _kernels(out.read_text()):
        fam = next((f for f in families if f in name), None)
        if fam is None:
            continue
        for loop in _dma_loops(body):
            seen[fam] += 1
            n = _drains_in_front_of_lds_reads(loop)
            if n:
                offenders.append((name, n))
    assert all(seen.values()), f"no LDS-DMA + MFMA loop found in {[f for f, n in seen.items() if not n]}: the scan no longer matches the code"
    assert not offenders, f"compiler-inserted vmcnt(0) in front of LDS fragment reads: {offenders[:4]}"
    shutil.rmtree(tmp_path, ignore_errors=True)


def _asm(unit, tmp_path):
    out = tmp_path / (unit + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I" + CSRC, os.path.join(CSRC, unit), "-o", str(out)]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode == 0 and out.exists(), p.stderr[-2000:]
    return out.read_text()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fp8_prefill_mfma_is_the_unscaled_encoding(tmp_path):
    """gemm8.hip passes constant-zero scale operands to __builtin_amdgcn_mfma_scale_f32_*_f8f6f4 and relies on hipcc selecting the
    UNSCALED f8f6f4 encoding (hardware scale 1.0). A toolchain that emitted the scaled form with E8M0 code 0 (= 2^-127) would make every fp8
    prefill output collapse to ~0 and only the GPU tests would notice: pin the instruction selection at build time."""
    asm = _asm("gemm8.hip", tmp_path)
    n_f8 = len(re.findall(r"\bv_mfma_f32_(?:32x32x64|16x16x128)_f8f6f4\b", asm))
    assert n_f8 > 100, f"the fp8 prefill kernels no longer use the f8f6f4 MFMAs ({n_f8} found)"
    scaled = re.findall(r"\bv_mfma_scale_\w+|\bv_mfma_ld_scale\w*", asm)
    assert not scaled, f"hipcc selected the block-SCALED MFMA encoding {sorted(set(scaled))[:3]}: with zero scale operands that multiplies by 2^-127"
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit", ["attn.hip", "chain.hip", "xstat32.hip"])
def test_decode_kernels_hold_no_scratch(unit, tmp_path):
    """VERDICT r4 "next" 1c / 7: the decode-step kernels (decode_attention_k, attn_oproj16_k, decode_chain_k, xstat32_k / xsplit32_k) and
    rope_kv_prefill_k are built at their register caps (128 at four workgroups or sixteen waves per CU, 256 for the activation-stationary GEMMs)
    and must not spill: a spilled dword comes back through scratch behind an `s_waitcnt vmcnt(0)` that drains the wave's whole K / V / weight
    window (round 4 carried 8-20 bytes in three of them)."""
    asm = _asm(unit, tmp_path)
    sizes = re.findall(r"\.amdhsa_kernel (\S+)|\.amdhsa_private_segment_fixed_size (\d+)", asm)
    cur, bad, n = None, [], 0
    for k, sz in sizes:
        if k:
            cur = k
        elif sz:
            n += 1
            if int(sz):
                bad.append((cur, int(sz)))
    assert n >= 4, "no kernel descriptors found: the scan no longer matches hipcc's assembly"
    assert not bad, f"kernels with a private (scratch) segment: {bad}"
    assert "scratch_" not in asm
    shutil.rmtree(tmp_path, ignore_errors=True)

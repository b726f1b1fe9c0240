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

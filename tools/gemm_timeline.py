"""In-kernel timeline of gemm_dma_k for one GEMM shape: python tools/gemm_timeline.py M N K [epi]"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import sys, torch
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
M, N, K = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = RdxEngine(small_cfg(), dtype="bf16", device=0, max_batch=1, max_len=32, llama=False, vision=False)
W = 2048
ms, tr = eng.kernel_bench(M, N, K, 0, 0, 1, epi, 10, trace_wgs=W)
tr = tr[tr[:, 0] > 0].double()
t0 = tr[:, 0].min()
us = lambda c: (c - t0) / 100.0
print(f"M={M} N={N} K={K} epi={epi}: {ms*1e3:.1f} us per launch, {tr.shape[0]} workgroups traced (last launch), steps/WG {int(tr[0,4])}")
print("  entry  us: min %.2f  med %.2f  max %.2f" % (us(tr[:, 0]).min(), us(tr[:, 0]).median(), us(tr[:, 0]).max()))
print("  first stage landed - entry: med %.2f max %.2f" % ((tr[:, 1] - tr[:, 0]).median() / 100, (tr[:, 1] - tr[:, 0]).max() / 100))
print("  k loop (after first stage): med %.2f max %.2f  -> per step %.3f us" % ((tr[:, 2] - tr[:, 1]).median() / 100, (tr[:, 2] - tr[:, 1]).max() / 100, (tr[:, 2] - tr[:, 1]).median() / 100 / max(tr[0, 4] - 1, 1)))
print("  epilogue: med %.2f max %.2f" % ((tr[:, 3] - tr[:, 2]).median() / 100, (tr[:, 3] - tr[:, 2]).max() / 100))
print("  end us: med %.2f max %.2f" % (us(tr[:, 3]).median(), us(tr[:, 3]).max()))
clk = ((tr[:, 7] - tr[:, 6]) / ((tr[:, 3] - tr[:, 0]) / 100.0)).median()       # shader cycles (s_memtime) per us of s_memrealtime
print("  shader clock while the kernel runs: %.0f MHz" % clk)
eng.close()

"""In-kernel timeline of wstat_k (the single prompt's weight-stationary GEMM) for one shape:  python tools/wstat_trace.py M N K [epi]
Per workgroup: entry, first row tile's K loop done (= weights + first ring landed), last row tile begins, end (100 MHz ticks)."""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import os, sys
os.environ["RDX_KB_WSTAT"] = "1"
import torch
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
M, N, K = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = RdxEngine(small_cfg(), dtype="bf16", device=0, max_batch=1, max_len=32, llama=False, vision=False)
ms, tr = eng.kernel_bench(M, N, K, 0, 0, 1, epi, 10, trace_wgs=2048)
tr = tr[tr[:, 0] > 0].double()
t0 = tr[:, 0].min()
us = lambda c: (c - t0) / 100.0
mt = (M + 15) // 16
print(f"M={M} N={N} K={K} epi={epi}: {ms*1e3:.1f} us per launch, {tr.shape[0]} workgroups (last launch), {mt} row tiles")
q = lambda v: "min %.2f med %.2f max %.2f" % (v.min(), v.median(), v.max())
print("  entry us:", q(us(tr[:, 0])))
print("  weights + first ring landed, first K loop done - entry:", q((tr[:, 1] - tr[:, 0]) / 100))
print("  row tiles 1 .. MT-2 (per tile):", q((tr[:, 2] - tr[:, 1]) / 100 / max(mt - 2, 1)))
print("  last tile + epilogue:", q((tr[:, 3] - tr[:, 2]) / 100))
if mt > 3:
    print("  row tile 4, wave 0: K loop", q((tr[:, 5] - tr[:, 4]) / 100), "| barrier", q((tr[:, 6] - tr[:, 5]) / 100), "| epilogue", q((tr[:, 7] - tr[:, 6]) / 100))
print("  workgroup lifetime:", q((tr[:, 3] - tr[:, 0]) / 100))
print("  end us:", q(us(tr[:, 3])))
eng.close()

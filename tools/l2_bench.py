"""Per-CU operand bandwidth from L2 / MALL (rdx_l2_bench): python tools/l2_bench.py"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import ctypes as C
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
from radialog_amd._lib import check
eng = RdxEngine(small_cfg(), dtype="bf16", device=0, max_batch=1, max_len=32, llama=False, vision=False)
def run(mode, bytes_per_wg, shared, reps, wgs):
    g = C.c_float(0)
    check(eng.ctx, eng.lib.rdx_l2_bench(eng.ctx, mode, bytes_per_wg, shared, reps, wgs, C.byref(g)), "rdx_l2_bench")
    return g.value
for mode, nm in ((0, "global_load_dwordx4 -> VGPR"), (1, "global_load_lds_dwordx4 (LDS-DMA)")):
    for wgs in (64, 144, 256, 512, 1024):
        for per, shared, what in ((1 << 16, 0, "64 KiB own region (L2)"), (1 << 20, 1, "1 MiB shared by all (L2)"), (1 << 20, 0, "1 MiB own region (L2/MALL)")):
            g = run(mode, per, shared, 200 if per <= (1 << 16) else 20, wgs)
            print(f"{nm:34s} {wgs:5d} WGs  {what:28s} {g/1e3:7.2f} TB/s aggregate = {g/min(wgs,256):7.1f} GB/s per CU")
eng.close()

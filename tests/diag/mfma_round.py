"""Diagnostic: how often does a 4096-deep GEMM output, rounded to the model dtype, differ from the exactly-accumulated (fp64)
value rounded the same way -- for the HIP kernels (MFMA fp32 accumulation) and for torch's CPU linear (fp32 FMA chain)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radialog_amd import synth
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine

for dtype, tdt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    eng = RdxEngine(small_cfg(), dtype=dtype, device=0, max_batch=2, max_len=64, vision=False)
    for K in (4096, 11008):
        N = 4096
        w = synth.synth(f"d.w{K}", (N, K), -0.035, 0.035)
        wq = w.to(tdt)
        for M in (8, 64, 160):
            x = synth.synth(f"d.x{K}.{M}", (M, K), -2.0, 2.0).to(tdt)
            exact = (x.double() @ wq.double().T)
            ex_r = exact.to(tdt)
            cpu = torch.nn.functional.linear(x, wq)
            f32 = (x.float() @ wq.float().T).to(tdt)
            hip = eng.gemm_test(x, wq.float(), epi=0).cpu()
            tot = ex_r.numel()
            fr = lambda a: float((a != ex_r).sum()) / tot
            # signed bias of the fp32 sums is invisible after rounding; look at the direction of the flips instead
            d = (hip.double() - ex_r.double())
            toward_zero = float(((d != 0) & (d.sign() != exact.sign())).sum()) / max(float((d != 0).sum()), 1)
            print(f"{dtype} K={K} M={M}: outputs != round(exact): hip {fr(hip):.4%}  torch-cpu {fr(cpu):.4%}  fp32-matmul {fr(f32):.4%}; "
                  f"hip flips toward zero: {toward_zero:.1%}")
    eng.close()

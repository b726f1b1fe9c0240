import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from radialog_amd import synth
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
for dtype, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
    e = RdxEngine(small_cfg(), dtype=dtype, device=0, max_batch=1, max_len=64, vision=False, llama=False)
    M, N, K = 1, 16, 64
    x = torch.ones(M, K).to(dt)
    w = synth.synth("dbg.w", (N, K), -0.05, 0.05)
    absmax = w.abs().amax(1, keepdim=True); q = (w * (448.0 / absmax)).to(torch.float8_e4m3fn).float() * (absmax / 448.0)
    out = e.gemm_test(x, w, None, None, 0, None, 1e-6, 4).float().cpu()
    print(dtype, "gpu", out[0, :6].tolist(), "ref", (x.float() @ q.t())[0, :6].tolist())
    out2 = e.gemm_test(x, w, None, None, 0, None, 1e-6, 0).float().cpu()
    print(dtype, "plain gpu", out2[0, :6].tolist(), "ref", (x.float() @ w.to(dt).float().t())[0, :6].tolist())
    e.close()

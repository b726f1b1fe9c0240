"""Does a finer activation scale reduce the e4m3 x e4m3 GEMM error? (CPU, torch.float8_e4m3fn; python tools/fp8_scale_study.py)
y = x W^T with W e4m3 per output row (the library's rule) and x quantised three ways: one absmax / 448 scale per row (the library, G = 1),
one per 32-element block (fp32 absmax / 448), one E8M0 (power-of-two) scale per 32-element block (the MX format the scaled MFMA takes).
Activations: unit normal, and the same with 8 outlier channels of 100 x the typical magnitude (what trained LLaMA residual streams show)."""
import torch

torch.manual_seed(0)
E4 = torch.float8_e4m3fn


def q_rows(v, block):
    shp = v.shape
    b = v.reshape(shp[0], -1, block)
    amax = b.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    return ((b * (448.0 / amax)).to(E4).float() * (amax / 448.0)).reshape(shp)


def q_mx(v, block=32):
    shp = v.shape
    b = v.reshape(shp[0], -1, block)
    amax = b.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0))                      # smallest power of two that keeps the block inside +-448
    sc = torch.exp2(e)
    return ((b / sc).to(E4).float() * sc).reshape(shp)


def study(name, x, w):
    wq = q_rows(w, w.shape[1])
    ref = x.double() @ w.double().T
    refq = x.double() @ wq.double().T                              # weight quantisation alone
    rows = [("weights only (x exact)", refq)]
    for label, xq in (("x: one scale per row (library)", q_rows(x, x.shape[1])), ("x: fp32 scale per 32 block", q_rows(x, 32)), ("x: E8M0 scale per 32 block (MX)", q_mx(x))):
        rows.append((label, xq.double() @ wq.double().T))
    rms = ref.pow(2).mean().sqrt()
    print(f"{name}: |y| rms {rms:.3f}")
    for label, y in rows:
        err = (y - ref)
        print(f"   {label:36s} rms error / rms(y) = {float(err.pow(2).mean().sqrt() / rms):.4f}   worst |error| / rms(y) = {float(err.abs().max() / rms):.3f}")


M, K, N = 32, 4096, 4096
w = (torch.rand(N, K) - 0.5) * 0.1
x = torch.randn(M, K)
study("unit-normal activations", x, w)
xo = x.clone()
xo[:, torch.randperm(K)[:8]] *= 100.0
study("8 outlier channels x 100", xo, w)
xs = x * torch.exp(torch.randn(1, K) * 1.5)                        # log-normal per-channel magnitudes (sigma 1.5: 20 x spread)
study("log-normal channel magnitudes", xs, w)

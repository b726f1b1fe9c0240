"""Yardstick only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) needs for the prefill shapes.

    python tools/blas_yardstick.py [M ...]

Prints, per (M, N, K) of the Vicuna-7B projections, the average time of `x @ w.T` in bf16 with both operands L2-cold between calls
(a rotating set of weight copies larger than the Infinity Cache), next to bytes / 6.3 TB/s and flops / 2.5 PFLOP/s.
"""
import sys

import torch


def main():
    ms = [int(a) for a in sys.argv[1:]] or [160, 256, 5120]
    dev = torch.device("cuda:0")
    shapes = [("qkv", 12304, 4096), ("o", 4096, 4096), ("gate/up", 22016, 4096), ("down", 4096, 11008)]
    for M in ms:
        for name, N, K in shapes:
            copies = max(2, int(600e6 // (N * K * 2)) + 1)
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            for i in range(3):
                (x @ ws[i % copies].t())
            torch.cuda.synchronize()
            iters = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                (x @ ws[i % copies].t())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            byt = (N * K + M * K + M * N) * 2
            fl = 2.0 * M * N * K
            print(f"M={M:5d} {name:8s} N={N:6d} K={K:6d}: {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s  {byt / us / 1e3:7.0f} GB/s   "
                  f"floors: hbm {byt / 6.3e6:6.1f} us, mfma {fl / 2.5e9:6.1f} us", flush=True)
            del ws


if __name__ == "__main__":
    main()

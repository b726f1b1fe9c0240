"""Per-shape timing of the encoder's GEMMs / convolutions through the production dispatch (rdx_kernel_bench):
python tools/enc_kernels.py [batch]   -- prints us, TFLOP/s and GB/s (algorithmic bytes: in + out (+ residual) + weights)"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import sys
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
only = sys.argv[2] if len(sys.argv) > 2 else ""
eng = RdxEngine(small_cfg(), dtype="bf16", device=0, max_batch=1, max_len=32, llama=False, vision=False)
RELU, NONE, RESRELU, RESID, GELU = 1, 0, 6, 3, 2
shapes = []
def conv(name, H, cin, cout, k, stride, epi): shapes.append((name, ("conv", H, cin, cout, k, stride, epi)))
def gemm(name, M, N, K, epi): shapes.append((name, ("gemm", M, N, K, epi)))
conv("l1.c1a 64->64", 112, 64, 64, 1, 1, RELU); conv("l1.c2 3x3 64", 112, 64, 64, 3, 1, RELU); conv("l1.c3 64->256 +res", 112, 64, 256, 1, 1, RESRELU)
conv("l1.ds 64->256", 112, 64, 256, 1, 1, NONE); conv("l1.c1b 256->64", 112, 256, 64, 1, 1, RELU)
conv("l2.c1a 256->128", 112, 256, 128, 1, 1, RELU); conv("l2.c2a 3x3 s2", 112, 128, 128, 3, 2, RELU); conv("l2.ds 256->512 s2", 112, 256, 512, 1, 2, NONE)
conv("l2.c3 128->512 +res", 56, 128, 512, 1, 1, RESRELU); conv("l2.c1b 512->128", 56, 512, 128, 1, 1, RELU); conv("l2.c2 3x3 128", 56, 128, 128, 3, 1, RELU)
conv("l3.c1a 512->256", 56, 512, 256, 1, 1, RELU); conv("l3.c2a 3x3 s2", 56, 256, 256, 3, 2, RELU); conv("l3.ds 512->1024 s2", 56, 512, 1024, 1, 2, NONE)
conv("l3.c3 256->1024 +res", 28, 256, 1024, 1, 1, RESRELU); conv("l3.c1b 1024->256", 28, 1024, 256, 1, 1, RELU); conv("l3.c2 3x3 256", 28, 256, 256, 3, 1, RELU)
conv("l4.c1a 1024->512", 28, 1024, 512, 1, 1, RELU); conv("l4.c2a 3x3 s2", 28, 512, 512, 3, 2, RELU); conv("l4.ds 1024->2048 s2", 28, 1024, 2048, 1, 2, NONE)
conv("l4.c3 512->2048 +res", 14, 512, 2048, 1, 1, RESRELU); conv("l4.c1b 2048->512", 14, 2048, 512, 1, 1, RELU); conv("l4.c2 3x3 512", 14, 512, 512, 3, 1, RELU)
P = 196
gemm("b2v 2048->256", B * P, 256, 2048, NONE); gemm("proj1 256->1408", B * P, 1408, 256, RELU); gemm("proj2 1408->1408", B * P, 1408, 1408, NONE)
gemm("q.crossKV 1408->9216", B * P, 9216, 1408, NONE)
M = B * 32
gemm("q.qkv 768->2304", M, 2304, 768, NONE); gemm("q.o 768->768 +res", M, 768, 768, RESID); gemm("q.ffn1 768->3072 gelu", M, 3072, 768, GELU); gemm("q.ffn2 3072->768 +res", M, 768, 3072, RESID)
for name, sp in shapes:
    if only and only not in name: continue
    if sp[0] == "conv":
        _, H, cin, cout, k, stride, epi = sp
        Ho = (H + 2 * (k // 2) - k) // stride + 1
        ms = eng.kernel_bench(B, cout, cin, H, k, stride, epi, 10)
        Mo = B * Ho * Ho
        fl = 2.0 * Mo * cout * cin * k * k
        by = B * H * H * cin * 2 / (stride * stride if k == 1 else 1) + Mo * cout * 2 * (2 if epi in (3, 6) else 1) + cout * cin * k * k * 2
    else:
        _, M_, N, K, epi = sp
        ms = eng.kernel_bench(M_, N, K, 0, 0, 1, epi, 10)
        fl = 2.0 * M_ * N * K
        by = M_ * K * 2 + M_ * N * 2 * (2 if epi in (3, 6) else 1) + N * K * 2
    print(f"{name:26s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  {by/ms/1e6:7.0f} GB/s   floor {max(fl/2.5e15, by/6.3e12)*1e6:6.1f} us")
eng.close()

/* rdx_hooks.h -- kernel-test, trace and microbenchmark hooks of librdx: a SEPARATE shared library, radialog_amd/librdx_hooks.so, built from
 * radialog_amd/csrc/api_debug.hip and linked against librdx.so. None of these is on the product path (the reference has no counterpart for any
 * of them); the parity tests and the tools/ scripts drive single kernels through them. radialog_amd/_lib.py loads the library only when
 * RDX_DEBUG_HOOKS=1 is set (tests/conftest.py and the tools set it); without it the Python wrappers raise RdxLibraryError. */
#ifndef RDX_HOOKS_H
#define RDX_HOOKS_H
#include "rdx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* debug: 8 timestamps (100 MHz ticks) of workgroup (0,0) of the stand-alone decode-attention kernel of `layer` */
int rdx_attn_trace(rdx_ctx* ctx, int layer, long long* host);
/* the fp8 path's activation quantisers on caller data (round 6: their e4m3 CODES and scales are compared with the oracle's fake quantisation, bit for bit).
 * X [M][K] model dtype (device), K % 128 == 0. mode 0: quant_rows_k -- out8 [M][K] e4m3 bytes, scales [M][groups] (groups 1..4: 128-deep blocks
 * [NB q / G, NB (q + 1) / G)); mode 1: RMSNorm(norm_w [K], eps) -> e4m3 -- out8 [M][K], scales [M]. */
int rdx_quant_test(rdx_ctx* ctx, const void* X, int M, int K, int groups, int mode, const void* norm_w, float eps, void* out8, float* scales);
/* debug: per-workgroup timestamps of one stand-alone decode GEMV (what: 1 gate/up, 2 qkv, 4 down), host[tile*8 + 0..5]; the
 * persistent batch >= 3 kernels (xstat32.hip) write one record per WORKGROUP: entry, first trip's K loop done, first trip done, last
 * trip begins, its K loop done, end, [6] = trips, [7] = XCC id (tools/xs_trace.py) */
int rdx_gemv_trace(rdx_ctx* ctx, int what, int layer, long long* host, int max_tiles);

/* one bare GEMM through the production kernels: out = epilogue(X . W^T); X/resid/norm_w/out model dtype, W [N][K] and
 * bias fp32. epi: 0 none, 1 relu, 2 gelu, 3 +resid, 4 swiglu(interleaved gate/up rows), 6 relu(+resid).
 * force: 0 = production dispatch (M <= 32 -> skinny, else LDS-DMA GEMM when K % 64 == 0, else tiled), 1 = skinny,
 * 2 = tiled_gemm_k, 3 = gemm_dma_k, 4 = skinny with fp8 (e4m3 + per-row scale) weights, 5 = the batch 3-32 K-split path (epi 3 only:
 * pack X, xsplit32_k, slab combine + residual), 6 = 5 with fp8 weights, 7 = the encoder's many-row kernel wsgemm_k (K % 64 == 0, epi 0-3 / 6;
 * RDX_WS_CFG=A..E forces a tile shape), 8 = the single prompt's weight-stationary kernel wstat_k (K = 4096 / 11008, epi 0 / 3 / 4: the rows are
 * RMS-normalised or just re-laid into the fragment-packed order by rmsnorm_k<T, 3>, then streamed past the register-resident weights),
 * 9 / 10 / 11 = the fp8 path's prefill GEMM gemm8 (e4m3 weights x e4m3 activations with 1 / 2 / 4 K groups; K % 64 == 0, epi 0 / 3 / 4).
 * force 4 / 6 hold the fp8 weights as the engine does (e4m3 bytes + scales only): batch >= 3 shapes multiply fp8 x fp8.
 * Test / benchmark hook. */
int rdx_gemm_test(rdx_ctx* ctx, const void* X, const float* W, const float* bias, const void* resid, void* out, int M, int N,
                  int K, int epi, const void* norm_w, float eps, int force);

/* kernel benchmark hook: ms per launch of one GEMM (ksize 0: rows x K -> N) or one NHWC convolution (ksize 1 / 3: `rows` images of
 * H x H x K channels -> N channels, stride, pad ksize / 2) through the dispatch the encoder / prefill use; zero-filled operands. */
int rdx_kernel_bench(rdx_ctx* ctx, int rows, int N, int K, int H, int ksize, int stride, int epi, int iters, float* ms_host,
                     long long* trace_host /* nullable: [trace_wgs][8] per-workgroup timestamps of gemm_dma_k (plain GEMMs) */, int trace_wgs);

/* one NHWC convolution (ksize 1 | 3, pad ksize / 2) on caller data: path 0 = the production dispatch of the row-major kernels, 1 = the fragment-packed
 * family pconv_k (round 4; pack -> conv -> unpack), 2 = pconv_k writing row-major itself. X [B][H][H][Cin], resid / out [B][Ho][Ho][Cout] model dtype;
 * W [Cout][ksize^2 Cin] fp32 with K ordered (kh, kw, c); ms_host (nullable): ms per launch of the convolution kernel alone. Test / benchmark hook
 * (the reference's convolutions are torchvision's, behind biovil_t/resnet.py:25-47). */
int rdx_conv_test(rdx_ctx* ctx, const void* X, const float* W, const float* bias, const void* resid, void* out, int B, int H, int Cin, int Cout,
                  int ksize, int stride, int epi, int path, int iters, float* ms_host);

/* microbenchmark: aggregate GB/s that `wgs` 256-thread workgroups pull from a cache-resident buffer (bytes_per_wg each, read `reps`
 * times; shared = 1: all read the same region); mode 0 = global_load_dwordx4 to registers, 1 = global_load_lds_dwordx4 (LDS-DMA) */
int rdx_l2_bench(rdx_ctx* ctx, int mode, long long bytes_per_wg, int shared, int reps, int wgs, float* gbps_host);

/* the lm_head epilogue of the weight-streaming kernels on a bare GEMM (M <= 32): logits model-dtype [M][N] (columns >=
 * n_valid are not written) and the greedy choice per row (argmax over n < n_valid, ties -> lowest index). Test hook. */
int rdx_logits_test(rdx_ctx* ctx, const void* X, const float* W, int M, int N, int n_valid, int K, void* out_logits,
                    int32_t* argmax_host, int fp8);

#ifdef __cplusplus
}
#endif
#endif /* RDX_HOOKS_H */

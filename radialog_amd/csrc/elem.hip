// Normalisation, layout and bookkeeping kernels (HBM-bound, vectorised where the layout allows).
#include <algorithm>
#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

// ---- e4m3 activation quantisation of the fp8 path (gemm8.hip, xstat32.hip): q = RNE_e4m3(x * (448 / absmax)), scale = absmax / 448 ----------
// (the arithmetic of pack_weight_fp8_k and of the oracle's fake quantisation: an all-zero range gets scale 1)
// where the 8 fp8 bytes of elements i .. i + 8 of row `row` go. PACK 5: row-major [rows][H]. PACK 4: the 32-row block in the 64-deep fragment
// order xstat32_k<W8, A8> reads, [chunk j = i / 64][row tile mt][lane (g = (i % 64) / 16, r = row % 16)][16 bytes], this piece's half i & 8
template <int PACK> __device__ __forceinline__ unsigned char* dst8(unsigned char* out8, size_t row, int i, int H) {
    if (PACK == 5) return out8 + row * H + i;
    return out8 + ((size_t)(((i >> 6) * 2 + (int)(row >> 4)) * 64 + ((i & 63) >> 4) * 16 + (int)(row & 15)) << 4) + (i & 8);
}

// Row-wise e4m3 quantisation of model-dtype activations [rows][K] (row stride ldx) into row-major bytes [rows][K] + scales [rows][G]:
// K group q = 128-deep blocks [NB q / G, NB (q + 1) / G), NB = K / 128 (G <= 4; K % 128 == 0), one absmax / 448 scale per (row, group). One
// workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void quant_rows_k(const T* __restrict__ x, long ldx, unsigned char* __restrict__ out8, float* __restrict__ xscale,
                                                    int K, int G) {
    typedef typename Vec8<T>::type V8;
    __shared__ float red[32];
    __shared__ float gmax[4];
    const size_t row = blockIdx.x;
    const T* xr = x + row * ldx;
    const int NB = K >> 7;
    int end[4];                                                 // first element past group q
#pragma unroll
    for (int q = 0; q < 4; ++q) end[q] = q < G ? ((NB * (q + 1)) / G) * 128 : K;
    auto group_of = [&](int i) { return (i >= end[0]) + (i >= end[1]) + (i >= end[2]); };
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x * 8; i < K; i += blockDim.x * 8) {
        const float m = amax8<T>(as_vec8<T>(ldg16(xr + i)));
        const int gq = group_of(i);
#pragma unroll
        for (int q = 0; q < 4; ++q) mx[q] = (q == gq) ? fmaxf(mx[q], m) : mx[q];
    }
    for (int q = 0; q < G; ++q) {
        const float m = block_max(mx[q], red);
        if (threadIdx.x == 0) gmax[q] = m;
    }
    __syncthreads();
    if ((int)threadIdx.x < G) { float sc, inv; fp8_scale(gmax[threadIdx.x], sc, inv); xscale[row * G + threadIdx.x] = sc; }
    for (int i = threadIdx.x * 8; i < K; i += blockDim.x * 8) {
        float sc, inv;
        fp8_scale(gmax[group_of(i)], sc, inv);
        const u2 q = quant8<T>(as_vec8<T>(ldg16(xr + i)), inv);
        *reinterpret_cast<u2*>(out8 + row * K + i) = q;
    }
}

void launch_quant_rows(int dtype, const void* x, long ldx, void* out8, float* xscale, int rows, int K, int groups, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((quant_rows_k<T>), dim3(rows), dim3(256), 0, s, (const T*)x, ldx, (unsigned char*)out8, xscale, K, groups));
}

// ---- LlamaRMSNorm (modeling_llama_imgemb.py:85-93): fp32 statistics, (x*rstd).to(T), then weight*h in T ---------------
// PACK != 0 (batch-32 decode, consumer xstat32_k): the output is written in the MFMA B-operand fragment order of the 32-row
// block, [fragment f][row tile mt][lane (g, r)][8], so that a wave's activation fragment is one contiguous KiB. A thread's 8
// elements k = i .. i + 8 of row m are exactly one lane's piece: PACK 1 (32-deep fragments): f = i / 32, g = (i % 32) / 8;
// PACK 2 (the fp8 weights' 64-deep chunks): f = 2 (i / 64) + (i % 16) / 8, g = (i % 64) / 16. Rows >= n_rows are zero-filled.
// PACK 3 (one prompt's prefill, consumer wstat_k): the PACK 1 order over `groups` row tiles instead of 2 (no slabs on that path).
// PACK 4 / 5 (fp8 path): the normalised row is quantised to e4m3 with ONE absmax / 448 scale (xscale[row]); 5 = row-major bytes (prefill,
// gemm8.hip), 4 = the 32-row block in the 64-deep fragment order (batch 3-32 decode, xstat32_k<W8, A8>); rows >= n_rows are zero bytes, scale 1.
template <typename T, int PACK>
__global__ __launch_bounds__(256) void rmsnorm_k(const T* x, const T* __restrict__ w, T* __restrict__ out,
                                                 int H, float eps, int n_rows, const float* __restrict__ slab, int groups, T* xw,
                                                 float* __restrict__ xscale = nullptr) {
    typedef typename Vec8<T>::type V8;
    __shared__ float red[32];
    const size_t row = blockIdx.x;
    auto dst = [&](int i) -> T* {
        if (PACK == 0) return out + row * H + i;
        const int f = PACK != 2 ? (i >> 5) : (2 * (i >> 6) + ((i & 15) >> 3)), g = PACK != 2 ? ((i & 31) >> 3) : ((i & 63) >> 4);
        return out + ((size_t)((f * (PACK == 3 ? groups : 2) + (int)(row >> 4)) * 64 + g * 16 + (int)(row & 15)) << 3);
    };
    if (PACK >= 4 && (int)row >= n_rows) {
        for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) *reinterpret_cast<u2*>(dst8<PACK == 4 ? 4 : 5>(reinterpret_cast<unsigned char*>(out), row, i, H)) = (u2){0u, 0u};
        if (threadIdx.x == 0) xscale[row] = 1.0f;
        return;
    }
    if (PACK && PACK < 4 && (int)row >= n_rows) {
        for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) stg16(dst(i), (u4){0u, 0u, 0u, 0u});
        return;
    }
    const T* xr = x + row * H;
    float ss = 0.f;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        V8 v = as_vec8<T>(ldg16(xr + i));
        if (PACK != 3 && slab) {
            // pending K-split projection (xsplit32_k): x[row] += T(sum of the groups' fp32 partials, fixed order) -- the residual
            // epilogue of o_proj / down_proj, done here at the launch boundary; the completed row is written back (same thread
            // re-reads it below)
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int gq = 0; gq < groups; ++gq) {
                const float* sp = slab + ((size_t)gq * 32 + row) * H + i;
                const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                acc[0] += s0.x; acc[1] += s0.y; acc[2] += s0.z; acc[3] += s0.w;
                acc[4] += s1.x; acc[5] += s1.y; acc[6] += s1.z; acc[7] += s1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fromf<T>(tof<T>(v[j]) + rnd<T>(acc[j]));
            stg16(xw + row * H + i, as_u4<T>(v));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = tof<T>(v[j]); ss += f * f; }
    }
    ss = block_sum(ss, red);
    const float rs = rsqrtf(ss / (float)H + eps);
    auto normed = [&](int i) -> V8 {
        V8 v = as_vec8<T>(ldg16(xr + i));
        if (!w) return v;                           // w == null: re-layout only (test hook)
        V8 o;
        V8 wv = as_vec8<T>(ldg16(w + i));
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof<T>(wv[j]) * rnd<T>(tof<T>(v[j]) * rs));
        return o;
    };
    if (PACK >= 4) {
        float am = 0.f;
        for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) am = fmaxf(am, amax8<T>(normed(i)));
        am = block_max(am, red);
        float sc, inv;
        fp8_scale(am, sc, inv);
        if (threadIdx.x == 0) xscale[row] = sc;
        for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8)
            *reinterpret_cast<u2*>(dst8<PACK == 4 ? 4 : 5>(reinterpret_cast<unsigned char*>(out), row, i, H)) = quant8<T>(normed(i), inv);
        return;
    }
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) stg16(dst(i), as_u4<T>(normed(i)));
}



void launch_rmsnorm_packed(int dtype, const void* x, const void* w, void* out, int rows, int mtiles, int H, float eps, hipStream_t s);

// One-round-trip version for H = 4096 (decode at batch 3-32: two of these per layer sit on the step's critical path): a
// thread owns 2 x 8 elements, every load (row, slabs, norm weight) is issued up front, the row stays in registers between the
// statistics and the scaling. Same arithmetic and rounding points as rmsnorm_k.
template <typename T, int PACK>
__global__ __launch_bounds__(256) void rmsnorm4096_k(const T* x, const T* __restrict__ w, T* __restrict__ out, float eps, int n_rows,
                                                     const float* __restrict__ slab, int groups, T* xw, float* __restrict__ xscale = nullptr, int mtl = 0) {
    // PACK 6 (round 5, 33-128 decoder rows): the PACK 3 order over `mtl` row tiles WITH the pending K-split slabs [groups][16 mtl][H] folded in first
    // PACK 4 with mtl > 0 (fp8 at 33-128 rows): ceil(mtl / 2) blocks of 32 rows, block b = the PACK 4 e4m3 block at byte offset 32 H b; slabs [groups][32 blocks][H]
    typedef typename Vec8<T>::type V8;
    constexpr int H = 4096;
    __shared__ float red[32];
    const size_t row = blockIdx.x;
    auto dst = [&](int i) -> T* {
        if (PACK == 0) return out + row * H + i;
        const int f = PACK != 2 ? (i >> 5) : (2 * (i >> 6) + ((i & 15) >> 3)), g = PACK != 2 ? ((i & 31) >> 3) : ((i & 63) >> 4);
        return out + ((size_t)((f * (PACK == 3 ? groups : PACK == 6 ? mtl : 2) + (int)(row >> 4)) * 64 + g * 16 + (int)(row & 15)) << 3);
    };
    const int i0 = threadIdx.x * 8, i1 = i0 + 2048;
    const size_t row8 = PACK == 4 ? (row & 31) : row;          // PACK 4: the row inside its 32-row e4m3 block ...
    unsigned char* const o8 = reinterpret_cast<unsigned char*>(out) + (PACK == 4 ? (row >> 5) * (size_t)(32 * H) : (size_t)0);      // ... and the block
    if ((PACK == 4 || PACK == 5) && (int)row >= n_rows) {
        *reinterpret_cast<u2*>(dst8<PACK == 4 ? 4 : 5>(o8, row8, i0, H)) = (u2){0u, 0u};
        *reinterpret_cast<u2*>(dst8<PACK == 4 ? 4 : 5>(o8, row8, i1, H)) = (u2){0u, 0u};
        if (threadIdx.x == 0) xscale[row] = 1.0f;
        return;
    }
    if (PACK && (PACK < 4 || PACK == 6) && (int)row >= n_rows) {
        stg16(dst(i0), (u4){0u, 0u, 0u, 0u});
        stg16(dst(i1), (u4){0u, 0u, 0u, 0u});
        return;
    }
    const T* xr = x + row * H;
    V8 v[2] = {as_vec8<T>(ldg16(xr + i0)), as_vec8<T>(ldg16(xr + i1))};
    const V8 wv[2] = {as_vec8<T>(ldg16(w + i0)), as_vec8<T>(ldg16(w + i1))};
    if (slab) {
        float4 sp[4][2][2];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float* p = slab + ((size_t)min(gq, groups - 1) * (PACK == 6 ? 16 * mtl : (PACK == 4 && mtl) ? 32 * ((mtl + 1) >> 1) : 32) + row) * H + (k ? i1 : i0);
                sp[gq][k][0] = *reinterpret_cast<const float4*>(p);
                sp[gq][k][1] = *reinterpret_cast<const float4*>(p + 4);
            }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                if (gq < groups) {
                    acc[0] += sp[gq][k][0].x; acc[1] += sp[gq][k][0].y; acc[2] += sp[gq][k][0].z; acc[3] += sp[gq][k][0].w;
                    acc[4] += sp[gq][k][1].x; acc[5] += sp[gq][k][1].y; acc[6] += sp[gq][k][1].z; acc[7] += sp[gq][k][1].w;
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = fromf<T>(tof<T>(v[k][j]) + rnd<T>(acc[j]));
            stg16(xw + row * H + (k ? i1 : i0), as_u4<T>(v[k]));
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = tof<T>(v[k][j]); ss += f * f; }
    ss = block_sum(ss, red);
    const float rs = rsqrtf(ss / (float)H + eps);
    V8 o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[k][j] = fromf<T>(tof<T>(wv[k][j]) * rnd<T>(tof<T>(v[k][j]) * rs));
    if (PACK == 4 || PACK == 5) {
        const float am = block_max(fmaxf(amax8<T>(o[0]), amax8<T>(o[1])), red);
        float sc, inv;
        fp8_scale(am, sc, inv);
        if (threadIdx.x == 0) xscale[row] = sc;
        *reinterpret_cast<u2*>(dst8<PACK == 4 ? 4 : 5>(o8, row8, i0, H)) = quant8<T>(o[0], inv);
        *reinterpret_cast<u2*>(dst8<PACK == 4 ? 4 : 5>(o8, row8, i1, H)) = quant8<T>(o[1], inv);
        return;
    }
    stg16(dst(i0), as_u4<T>(o[0]));
    stg16(dst(i1), as_u4<T>(o[1]));
}

void launch_rmsnorm_packed(int dtype, const void* x, const void* w, void* out, int rows, int mtiles, int H, float eps, hipStream_t s) {
    if (H == 4096 && w) {       // one-round-trip kernel, PACK 3: `groups` carries the row tiles
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm4096_k<T, 3>), dim3(mtiles * 16), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, eps, rows,
                                                    (const float*)nullptr, mtiles, (T*)nullptr));
        return;
    }
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm_k<T, 3>), dim3(mtiles * 16), dim3(256), 0, s, (const T*)x, (const T*)w,
                                                (T*)out, H, eps, rows, (const float*)nullptr, mtiles, (T*)nullptr));
}

// 33-128 decoder rows: x += T(sum of the `groups` pending K-split slabs [groups][16 mtiles][H]) (written back), then RMSNorm into the fragment-packed
// [k / 32][mtiles][lane][8] the row-block kernels read. H = 4096.
void launch_rmsnorm_packed_slab(int dtype, void* x, const void* w, void* out, int rows, int mtiles, float eps, const float* slab, int groups, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm4096_k<T, 6>), dim3(mtiles * 16), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, eps, rows, slab, groups,
                                                (T*)x, (float*)nullptr, mtiles));
}

void launch_rmsnorm(int dtype, const void* x, const void* w, void* out, int rows, int H, float eps, hipStream_t s) {
    if (H == 4096 && w) {       // the one-round-trip kernel (row kept in registers between the statistics and the scaling): same elements per thread, same order
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm4096_k<T, 0>), dim3(rows), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, eps, rows,
                                                    (const float*)nullptr, 0, (T*)nullptr));
        return;
    }
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm_k<T, 0>), dim3(rows), dim3(256), 0, s, (const T*)x, (const T*)w,
                                                (T*)out, H, eps, rows, (const float*)nullptr, 0, (T*)nullptr));
}

// RMSNorm -> e4m3 row-major bytes [rows][H] + one scale per row (the fp8 path's prefill: input of the QKV / gate-up gemm8 launches)
void launch_rmsnorm_fp8(int dtype, const void* x, const void* w, void* out8, float* xscale, int rows, int H, float eps, hipStream_t s) {
    if (H == 4096 && w) {
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm4096_k<T, 5>), dim3(rows), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out8, eps, rows,
                                                    (const float*)nullptr, 0, (T*)nullptr, xscale));
        return;
    }
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm_k<T, 5>), dim3(rows), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out8, H, eps, rows,
                                                (const float*)nullptr, 0, (T*)nullptr, xscale));
}

// ... and the batch 3-32 decode form: rows <= 32 into the fragment-packed 32-row fp8 block of xstat32_k<W8, A8> + xscale[32], with the pending
// K-split slabs folded in first like launch_rmsnorm_packed32
void launch_rmsnorm_packed32_fp8(int dtype, void* x, const void* w, void* out8, float* xscale, int rows, int H, float eps, const float* slab,
                                 int groups, hipStream_t s) {
    if (H == 4096 && w && groups <= 4) {
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm4096_k<T, 4>), dim3(32), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out8, eps, rows, slab,
                                                    groups, (T*)x, xscale));
        return;
    }
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm_k<T, 4>), dim3(32), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out8, H, eps, rows, slab,
                                                groups, (T*)x, xscale));
}

// ... and fp8 at 33-128 rows: ceil(mtiles / 2) such 32-row e4m3 blocks at a block stride of 32 H bytes, xscale[32 blocks], slabs [groups][32 blocks][H]. H = 4096.
void launch_rmsnorm_blk_fp8(int dtype, void* x, const void* w, void* out8, float* xscale, int rows, int mtiles, float eps, const float* slab, int groups, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rmsnorm4096_k<T, 4>), dim3(32 * ((mtiles + 1) / 2)), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out8, eps, rows, slab,
                                                groups, (T*)x, xscale, mtiles));
}

void launch_rmsnorm_packed32(int dtype, void* x, const void* w, void* out, int rows, int H, float eps, int pack, const float* slab,
                             int groups, hipStream_t s) {
    if (H == 4096 && w && groups <= 4) {
        RDX_DISPATCH_T(dtype, T, {
            if (pack == 2) hipLaunchKernelGGL((rmsnorm4096_k<T, 2>), dim3(32), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, eps, rows, slab, groups, (T*)x);
            else if (pack == 1) hipLaunchKernelGGL((rmsnorm4096_k<T, 1>), dim3(32), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, eps, rows, slab, groups, (T*)x);
            else hipLaunchKernelGGL((rmsnorm4096_k<T, 0>), dim3(rows), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, eps, rows, slab, groups, (T*)x);
        });
        return;
    }
    RDX_DISPATCH_T(dtype, T, {
        if (pack == 2) hipLaunchKernelGGL((rmsnorm_k<T, 2>), dim3(32), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, H, eps, rows, slab, groups, (T*)x);
        else if (pack == 1) hipLaunchKernelGGL((rmsnorm_k<T, 1>), dim3(32), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, H, eps, rows, slab, groups, (T*)x);
        else hipLaunchKernelGGL((rmsnorm_k<T, 0>), dim3(rows), dim3(256), 0, s, (const T*)x, (const T*)w, (T*)out, H, eps, rows, slab, groups, (T*)x);
    });
}

// ---- LayerNorm over the last dim (Q-Former post-LN, eps 1e-12; fp32 statistics, two-pass variance) -------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_k(const T* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ out,
                                                   float* __restrict__ out_f32, int H, float eps) {
    __shared__ float red[32];
    const size_t row = blockIdx.x;
    const T* xr = x + row * H;
    float s = 0.f;
    for (int i = threadIdx.x; i < H; i += blockDim.x) s += tof<T>(xr[i]);
    const float mean = block_sum(s, red) / (float)H;
    float v = 0.f;
    for (int i = threadIdx.x; i < H; i += blockDim.x) { const float d = tof<T>(xr[i]) - mean; v += d * d; }
    const float rstd = rsqrtf(block_sum(v, red) / (float)H + eps);
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        const float y = (tof<T>(xr[i]) - mean) * rstd * gamma[i] + beta[i];
        if (out) out[row * H + i] = fromf<T>(y);
        if (out_f32) out_f32[row * H + i] = y;
    }
}

void launch_layernorm(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* out_f32, int rows,
                      int H, float eps, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((layernorm_k<T>), dim3(rows), dim3(256), 0, s, (const T*)x, gamma, beta,
                                                (T*)out, out_f32, H, eps));
}

// ---- LayerNorm with row strides and an optional additive per-token embedding (ViT pooler: norm1(x) + pos/type emb) ---
template <typename T>
__global__ __launch_bounds__(256) void layernorm_ex_k(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const T* __restrict__ emb, int emb_rows,
                                                      T* __restrict__ out, long ldo, int H, float eps) {
    __shared__ float red[32];
    const size_t row = blockIdx.x;
    const T* xr = x + row * ldx;
    float s = 0.f;
    for (int i = threadIdx.x; i < H; i += blockDim.x) s += tof<T>(xr[i]);
    const float mean = block_sum(s, red) / (float)H;
    float v = 0.f;
    for (int i = threadIdx.x; i < H; i += blockDim.x) { const float d = tof<T>(xr[i]) - mean; v += d * d; }
    const float rstd = rsqrtf(block_sum(v, red) / (float)H + eps);
    const T* er = emb ? emb + (size_t)(row % emb_rows) * H : nullptr;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        float y = (tof<T>(xr[i]) - mean) * rstd * gamma[i] + beta[i];
        if (er) y += tof<T>(er[i]);
        out[row * ldo + i] = fromf<T>(y);
    }
}
void launch_layernorm_ex(int dtype, const void* x, long ldx, const float* gamma, const float* beta, const void* emb, int emb_rows,
                         void* out, long ldo, int rows, int H, float eps, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((layernorm_ex_k<T>), dim3(rows), dim3(256), 0, s, (const T*)x, ldx, gamma, beta,
                                                (const T*)emb, emb_rows, (T*)out, ldo, H, eps));
}

// ---- ViT pooler token plumbing: tokens[b][j] = (j < L ? current : previous) image's patch j % L; final concat ----------
template <typename T>
__global__ void pool_gather_k(const T* __restrict__ src, T* __restrict__ dst, int B, int L, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // one thread = 8 channels
    const int C8 = C >> 3;
    if (i >= (size_t)B * 2 * L * C8) return;
    const int c8 = (int)(i % C8);
    const size_t tok = i / C8;
    const int j = (int)(tok % (2 * L)), b = (int)(tok / (2 * L));
    const size_t srow = (size_t)(j < L ? b : B + b) * L + (j % L);
    stg16(dst + tok * C + c8 * 8, ldg16(src + srow * C + c8 * 8));
}
void launch_pool_gather(int dtype, const void* src, void* dst, int B, int L, int C, hipStream_t s) {
    const size_t n = (size_t)B * 2 * L * (C >> 3);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((pool_gather_k<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const T*)src,
                                                (T*)dst, B, L, C));
}
template <typename T>
__global__ void pool_concat_k(const T* __restrict__ patch, const T* __restrict__ tokens, T* __restrict__ dst, int B, int L, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // dst [B*L][2C]: [patch_x | diff_x]
    const int C8 = C >> 3;
    if (i >= (size_t)B * L * 2 * C8) return;
    const int c8 = (int)(i % (2 * C8));
    const size_t row = i / (2 * C8);
    const int b = (int)(row / L), j = (int)(row % L);
    const T* srcp = c8 < C8 ? patch + row * C + c8 * 8 : tokens + ((size_t)b * 2 * L + j) * C + (c8 - C8) * 8;
    stg16(dst + row * 2 * C + c8 * 8, ldg16(srcp));
}
void launch_pool_concat(int dtype, const void* patch, const void* tokens, void* dst, int B, int L, int C, hipStream_t s) {
    const size_t n = (size_t)B * L * 2 * (C >> 3);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((pool_concat_k<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const T*)patch,
                                                (const T*)tokens, (T*)dst, B, L, C));
}

// ---- the NCHW reshape scramble + ln_vision (blip2_qformer.py:469, blip2.py:199-205) ----------------------------------
// The projector output is produced NHWC ([B][P][C]); the reference reshapes its NCHW tensor [B][C][P] to [B][P][C] without a
// permute, so token row r, column c is flat element f = r*C + c of the [C][P] matrix: channel f / P, position f % P.
template <typename T>
__global__ __launch_bounds__(256) void scramble_layernorm_k(const T* __restrict__ pp, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ out,
                                                            float* __restrict__ out_f32, int P, int C, float eps) {
    __shared__ float red[32];
    extern __shared__ float rowbuf[];
    const int rrow = blockIdx.x, b = blockIdx.y;
    const T* src = pp + (size_t)b * P * C;
    float s = 0.f;
    if ((C & 7) == 0) {
        // the row's C elements are the flat range [rrow C, rrow C + C) of the [C][P] matrix: <= C / P + 2 channels x all P positions. Gathered as 16-byte
        // pieces (8 channels of one position) instead of one 2-byte element per 64-byte sector (round 3 PMC: 227 MB fetched for an 18 MB input at batch 32);
        // the sums below run over rowbuf in the order they always did, so the result is bit-identical.
        typedef typename Vec8<T>::type V8;
        const int f0 = rrow * C, ch_lo = (f0 / P) & ~7, ch_hi = (f0 + C - 1) / P;
        const int nch8 = ((ch_hi - ch_lo) >> 3) + 1;
        for (int it = threadIdx.x; it < P * nch8; it += blockDim.x) {
            const int pos = it / nch8, ch8 = ch_lo + (it - pos * nch8) * 8;
            if (ch8 >= C) continue;
            const V8 v = as_vec8<T>(ldg16(src + (size_t)pos * C + ch8));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = (ch8 + j) * P + pos - f0;
                if (c >= 0 && c < C) rowbuf[c] = tof<T>(v[j]);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) s += rowbuf[c];
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const size_t f = (size_t)rrow * C + c;
            const int ch = (int)(f / P), pos = (int)(f % P);
            const float v = tof<T>(src[(size_t)pos * C + ch]);
            rowbuf[c] = v;
            s += v;
        }
    }
    const float mean = block_sum(s, red) / (float)C;
    float v2 = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { const float d = rowbuf[c] - mean; v2 += d * d; }
    const float rstd = rsqrtf(block_sum(v2, red) / (float)C + eps);
    const size_t o = ((size_t)b * P + rrow) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float y = (rowbuf[c] - mean) * rstd * gamma[c] + beta[c];
        if (out) out[o + c] = fromf<T>(y);
        if (out_f32) out_f32[o + c] = y;
    }
}

void launch_scramble_layernorm(int dtype, const void* pp_nhwc, const float* gamma, const float* beta, void* out,
                               float* out_f32, int B, int P, int C, float eps, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((scramble_layernorm_k<T>), dim3(P, B), dim3(256), (size_t)C * sizeof(float), s,
                                                (const T*)pp_nhwc, gamma, beta, (T*)out, out_f32, P, C, eps));
}

// ---- image prep: float32 NCHW [B,3,S,S] -> T NHWC, zero-padded spatially, 4 channels (4th = 0) -------------------------
template <typename T>
__global__ __launch_bounds__(256) void img_prep_k(const float* __restrict__ img, T* __restrict__ out, int S, int pad, int Hp,
                                                  int Wp) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= Wp) return;
    typedef T T4 __attribute__((ext_vector_type(4)));
    T4 o;
    o[0] = o[1] = o[2] = o[3] = fromf<T>(0.f);
    const int iy = y - pad, ix = x - pad;
    if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
        const size_t plane = (size_t)S * S;
        const float* p = img + (size_t)b * 3 * plane + (size_t)iy * S + ix;
        o[0] = fromf<T>(p[0]); o[1] = fromf<T>(p[plane]); o[2] = fromf<T>(p[2 * plane]);
    }
    *reinterpret_cast<T4*>(out + (((size_t)b * Hp + y) * Wp + x) * 4) = o;
}

void launch_img_prep(int dtype, const float* img, void* out, int B, int S, int pad, int Hp, int Wp, hipStream_t s) {
    dim3 grid((Wp + 255) / 256, Hp, B), block(256);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((img_prep_k<T>), grid, block, 0, s, img, (T*)out, S, pad, Hp, Wp));
}

// ---- 3x3 stride-2 pad-1 max pool, NHWC, 8 channels per thread -------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_k(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C, int Ho,
                                                 int Wo, size_t total) {
    typedef typename Vec8<T>::type V8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int C8 = C >> 3;
    const int c8 = (int)(idx % C8);
    size_t rest = idx / C8;
    const int ow = (int)(rest % Wo); rest /= Wo;
    const int oh = (int)(rest % Ho);
    const int b = (int)(rest / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if (iw < 0 || iw >= W) continue;
            V8 v = as_vec8<T>(ldg16(in + (((size_t)b * H + ih) * W + iw) * C + c8 * 8));
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], tof<T>(v[j]));
        }
    }
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fromf<T>(m[j]);
    stg16(out + (((size_t)b * Ho + oh) * Wo + ow) * C + c8 * 8, as_u4<T>(o));
}

void launch_maxpool(int dtype, const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C >> 3);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((maxpool_k<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                                (const T*)in, (T*)out, H, W, C, Ho, Wo, total));
}

// ---- small copies / conversions --------------------------------------------------------------------------------------------
template <typename T>
__global__ void broadcast_rows_k(const T* __restrict__ src, T* __restrict__ dst, size_t per, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] = src[i % per];
}
void launch_broadcast_rows(int dtype, const void* src, void* dst, int rows, int H, int B, hipStream_t s) {
    const size_t per = (size_t)rows * H, total = per * B;
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((broadcast_rows_k<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                                (const T*)src, (T*)dst, per, total));
}

template <typename T> __global__ void to_f32_k(const T* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = tof<T>(src[i]);
}
template <typename T> __global__ void from_f32_k(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = fromf<T>(src[i]);
}
void launch_to_f32(int dtype, const void* src, float* dst, size_t n, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((to_f32_k<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const T*)src, dst, n));
}
void launch_from_f32(int dtype, const float* src, void* dst, size_t n, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((from_f32_k<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, (T*)dst, n));
}

// ---- prompt preparation: split_at_img position, attention mask, position ids (:498-520, :805-810) ---------------------
__global__ void prep_prompt_k(const int* __restrict__ ids, const int* __restrict__ mask_in, int T_, int img_id, int pad_id,
                              int* __restrict__ img_pos, int* __restrict__ pos_ids, uint8_t* __restrict__ key_mask, long km_bs,
                              int max_len_fill, int* __restrict__ pos_next, int* __restrict__ slot_b, int* __restrict__ step_b,
                              int* __restrict__ unfinished) {
    const int b = blockIdx.x;
    const int* row = ids + (size_t)b * T_;
    uint8_t* km = key_mask + (size_t)b * km_bs;
    if (threadIdx.x == 0) {
        int first = -1, cum = 0;
        for (int t = 0; t < T_; ++t) {
            if (first < 0 && row[t] == img_id) first = t;
            const int m = mask_in ? (mask_in[(size_t)b * T_ + t] != 0) : (row[t] != pad_id);
            cum += m;
            pos_ids[(size_t)b * T_ + t] = m ? (cum - 1) : 1;
            km[t] = (uint8_t)m;
        }
        img_pos[b] = first < 0 ? 0 : first;
        pos_next[b] = cum;        // position id of the first generated token = cumsum(mask) - 1 after appending a 1
        slot_b[b] = T_;
        step_b[b] = 0;
        unfinished[b] = 1;
    }
    for (int t = T_ + threadIdx.x; t < max_len_fill; t += blockDim.x) km[t] = 1;   // generated tokens are always attended
}
void launch_prep_prompt(const int* ids, const int* mask_in, int B, int T_, int img_id, int pad_id, int* img_pos, int* pos_ids,
                        uint8_t* key_mask, long km_bs, int* pos_next, int* slot_b, int* step_b, int* unfinished, hipStream_t s) {
    hipLaunchKernelGGL(prep_prompt_k, dim3(B), dim3(256), 0, s, ids, mask_in, T_, img_id, pad_id, img_pos, pos_ids, key_mask,
                       km_bs, (int)km_bs, pos_next, slot_b, step_b, unfinished);
}

// ---- continuation of a cached conversation: `T_` new prompt tokens go behind the first `keep_len` cache slots of every row
// (multi-turn re-prompting, test.py:440-674 / demo.py:277-305, without recomputing the shared prefix). A row's pad count is
// slot - next position, constant since its first prefill; the key mask of the kept slots stays, later slots are already 1.
__global__ void prep_append_k(int T_, int keep_len, int* __restrict__ img_pos, int* __restrict__ pos_ids, int* __restrict__ pos_next,
                              int* __restrict__ slot_b, int* __restrict__ step_b, int* __restrict__ unfinished) {
    const int b = blockIdx.x;
    const int pads = slot_b[b] - pos_next[b];
    const int base = keep_len - pads;
    for (int t = threadIdx.x; t < T_; t += blockDim.x) pos_ids[(size_t)b * T_ + t] = base + t;
    __syncthreads();
    if (threadIdx.x == 0) {
        img_pos[b] = 0;
        pos_next[b] = base + T_;
        slot_b[b] = keep_len + T_;
        step_b[b] = 0;
        unfinished[b] = 1;
    }
}
void launch_prep_append(int B, int T_, int keep_len, int* img_pos, int* pos_ids, int* pos_next, int* slot_b, int* step_b,
                        int* unfinished, hipStream_t s) {
    hipLaunchKernelGGL(prep_append_k, dim3(B), dim3(256), 0, s, T_, keep_len, img_pos, pos_ids, pos_next, slot_b, step_b, unfinished);
}

// ---- embedding gather + image splice (:571-594): rows [p, p+32) take the projected image embedding ----------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_k(const int* __restrict__ ids, const int* __restrict__ img_pos,
                                                      const T* __restrict__ embed, int vocab, const T* __restrict__ img_emb,
                                                      int n_img, T* __restrict__ out, int T_, int H, int use_img) {
    const int t = blockIdx.x, b = blockIdx.y;
    const T* src;
    const int p = img_pos[b];
    if (use_img && t >= p && t < p + n_img) {
        src = img_emb + ((size_t)b * n_img + (t - p)) * H;
    } else {
        int id = ids[(size_t)b * T_ + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = embed + (size_t)id * H;
    }
    T* dst = out + ((size_t)b * T_ + t) * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) stg16(dst + i, ldg16(src + i));
}
void launch_embed_splice(int dtype, const int* ids, const int* img_pos, const void* embed, int vocab, const void* img_emb,
                         int n_img, void* out, int B, int T_, int H, int use_img, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((embed_splice_k<T>), dim3(T_, B), dim3(256), 0, s, ids, img_pos, (const T*)embed,
                                                vocab, (const T*)img_emb, n_img, (T*)out, T_, H, use_img));
}

template <typename T>
__global__ void gather_last_k(const T* __restrict__ x, T* __restrict__ out, int T_, int H) {
    const int b = blockIdx.x;
    const T* src = x + ((size_t)b * T_ + (T_ - 1)) * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) stg16(out + (size_t)b * H + i, ldg16(src + i));
}
void launch_gather_last(int dtype, const void* x, void* out, int B, int T_, int H, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((gather_last_k<T>), dim3(B), dim3(256), 0, s, (const T*)x, (T*)out, T_, H));
}

// ---- greedy step (transformers 4.28.1 greedy_search rule) ----------------------------------------------------------------
// next = argmax(logits) (lowest index wins ties); finished rows emit pad; a row finishes when it emits eos.
// Also advances the per-row decode state and gathers the next input embedding, so the whole step stays on the device.
template <typename T>
__global__ __launch_bounds__(256) void greedy_step_k(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                     int n_tiles, int eos_id, int pad_id, int max_new, int* __restrict__ out_tokens,
                                                     int* __restrict__ unfinished, int* __restrict__ pos, int* __restrict__ slot_b,
                                                     int* __restrict__ step_b, const T* __restrict__ embed, int vocab,
                                                     T* __restrict__ x_next, int H, const int* pos_ro,
                                                     const T* __restrict__ cos_t, const T* __restrict__ sin_t, T* __restrict__ cur_rope,
                                                     int* __restrict__ ctr_zero, int n_zero) {
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ int tok_s, pos_s;
    const int b = blockIdx.x;
    // hand-off counters of the NEXT decode step's fused launches: zeroed here, by the last kernel of this step
    if (ctr_zero && b == 0) for (int i = threadIdx.x; i < n_zero; i += blockDim.x) ctr_zero[i] = 0;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) {
        const float v = part_val[(size_t)b * n_tiles + i];
        const int ix = part_idx[(size_t)b * n_tiles + i];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v = sv[threadIdx.x + o];
            const int ix = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = ix; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int tok = si[0];
        const int step = step_b[b];
        if (eos_id >= 0) {
            const int unf = unfinished[b];
            tok = unf ? tok : pad_id;
            if (tok == eos_id) unfinished[b] = 0;
        }
        if (step < max_new) out_tokens[(size_t)b * max_new + step] = tok;
        step_b[b] = step + 1;
        if (slot_b) slot_b[b] += 1;      // null on the prefill call: token 0 is consumed by the first decode step
        int pcur = pos_ro ? pos_ro[b] : 0;
        if (pos) { pcur = pos[b] + 1; pos[b] = pcur; }
        pos_s = pcur;
        tok_s = tok;
    }
    __syncthreads();
    int id = tok_s;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const T* src = embed + (size_t)id * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) stg16(x_next + (size_t)b * H + i, ldg16(src + i));
    // cos | sin row of the position the NEXT decode step works at, so that decode attention needs no pos -> table load chain
    if (cur_rope && threadIdx.x < 32) {
        const int half = threadIdx.x >> 4, c8 = (threadIdx.x & 15) * 8;
        const T* tab = half ? sin_t : cos_t;
        stg16(cur_rope + (size_t)b * 256 + half * 128 + c8, ldg16(tab + (size_t)pos_s * 128 + c8));
    }
}
// ---- avg_pool2d(pool) + NCHW flatten of the NHWC projector output (chexpert_model.py:17-18) ----------------------------
template <typename T>
__global__ void avgpool_flatten_k(const T* __restrict__ in, T* __restrict__ out, int B, int G, int C, int pool) {
    const int Gp = G / pool;                       // floor, like F.avg_pool2d
    const size_t total = (size_t)B * C * Gp * Gp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % Gp), h = (int)((i / Gp) % Gp), c = (int)((i / ((size_t)Gp * Gp)) % C), b = (int)(i / ((size_t)Gp * Gp * C));
        float acc = 0.f;
        for (int dy = 0; dy < pool; ++dy)
            for (int dx = 0; dx < pool; ++dx)
                acc += tof<T>(in[(((size_t)b * G + h * pool + dy) * G + w * pool + dx) * C + c]);
        out[i] = fromf<T>(acc / (float)(pool * pool));      // x.view(B, -1) of [B][C][Gp][Gp]
    }
}
void launch_avgpool_flatten(int dtype, const void* in, void* out, int B, int G, int C, int pool, hipStream_t s) {
    const size_t total = (size_t)B * C * (G / pool) * (G / pool);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((avgpool_flatten_k<T>), dim3(blocks), dim3(256), 0, s, (const T*)in, (T*)out, B, G, C, pool));
}

void launch_greedy_step(int dtype, const float* part_val, const int* part_idx, int n_tiles, int B, int eos_id, int pad_id,
                        int max_new, int* out_tokens, int* unfinished, int* pos, int* slot_b, int* step_b, const void* embed,
                        int vocab, void* x_next, int H, const int* pos_ro, const void* cos_t, const void* sin_t, void* cur_rope,
                        int* ctr_zero, int n_zero, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((greedy_step_k<T>), dim3(B), dim3(256), 0, s, part_val, part_idx, n_tiles, eos_id,
                                                pad_id, max_new, out_tokens, unfinished, pos, slot_b, step_b, (const T*)embed, vocab,
                                                (T*)x_next, H, pos_ro, (const T*)cos_t, (const T*)sin_t, (T*)cur_rope, ctr_zero, n_zero));
}

}  // namespace rdx

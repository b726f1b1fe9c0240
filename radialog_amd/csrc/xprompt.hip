// One prompt's K = 4096 projections (QKV, o_proj, gate/up of a 65-192 token prefill) on ROW BLOCKS OF 64 with the K range split in two (gfx950, round 5).
//
// What bounds a single prompt's projections is the chip's L2 -> CU delivery (13.4 TB/s measured): with the prompt cut into blocks of 32 rows
// (xstat32_k<.., BLK>) every weight fragment crosses the L2s ceil(M / 32) times -- 5 x at 160 tokens. The only lever is rows per weight pass. A workgroup
// cannot hold 64 rows x 4096 of activations (512 KiB), but it can hold 64 rows x HALF of K: here a (tile walker, row block of 64, K half) triple is one
// 8-wave workgroup -- wave w keeps the B fragments of its 4 row tiles x 8 chunks (k = 2048 kg + 256 w .. + 256) in 128 VGPRs, streams the matching halves of
// the weight tiles through a 16-fragment ring (two tiles per trip), does 4 MFMAs per weight fragment and leaves fp32 partial sums in a slab [K half][row][N].
// The U = 2 x ceil(M / 64) workgroups of a walker sit on ONE XCD (workgroup id % 8, as in xstat32_k<.., BLK>) and walk the same tiles: a weight fragment comes
// from HBM once and crosses that L2 ceil(M / 64) times -- 3 x at 160 tokens. The two K halves are added, rounded and finished (RoPE / residual + RMSNorm /
// SwiGLU) by the CONSUMER of each projection, in fixed order: rope_kv_prefill_k<.., SLAB>, rmsnorm4096_k<T, 6>, swiglu_slab_k below.
// Rounding points are the reference's (one rounding of the full fp32 sum to the model dtype); the accumulation order is (first K half) + (second K half).
#include <type_traits>
#include <algorithm>
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"   // swiglu()

namespace rdx {

constexpr int XP_WAVES = 8, XP_THREADS = 512, XP_K = 4096, XP_KH = 2, XP_MT = 4, XP_CPW = XP_K / 32 / XP_KH / XP_WAVES;   // 8 chunks per wave and K half
constexpr int XP_TPI = 2;
constexpr size_t XP_SMEM = (size_t)2 * XP_TPI * XP_WAVES * XP_MT * 256 * 4;      // [2 bufs][2 tiles][8 waves][4 row tiles][256] fp32 = 128 KiB

template <typename T>
__global__ __launch_bounds__(XP_THREADS) void xprompt64_k(GemmArgs a, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
    float* red = reinterpret_cast<float*>(smx);
    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = (a.N + 15) >> 4;
    // id = 8 slot + xcd; the 32 slots of an XCD = WPX walkers x (NB row blocks of 64 x 2 K halves)
    const int NB = (a.mtiles + XP_MT - 1) / XP_MT, U = NB * XP_KH, WPX = 32 / U;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    if (slot >= WPX * U) return;
    const int wid = xcd * WPX + slot / U, u = slot % U, mb = u / XP_KH, kg = u % XP_KH, G = 8 * WPX;
    const int ngroups = (ntiles + XP_TPI - 1) / XP_TPI;
    const int nit = (ngroups - wid + G - 1) / G;
    if (nit <= 0) return;

    const int cw0 = kg * (XP_K / 32 / XP_KH) + wa * XP_CPW;                         // this wave's first 32-deep chunk
    const u4* wbase = reinterpret_cast<const u4*>(a.W) + (size_t)cw0 * 64;         // wave-uniform
    auto tile_ptr = [&](int t) { return wbase + (size_t)min(t, ntiles - 1) * (XP_K / 32) * 64; };

    u4 ring[XP_TPI * XP_CPW];
#pragma unroll
    for (int q = 0; q < XP_TPI; ++q) {
        const u4* wp = tile_ptr(wid * XP_TPI + q);
#pragma unroll
        for (int j = 0; j < XP_CPW; ++j) {
            ring[q * XP_CPW + j] = ldg16_nt(wp + (unsigned)(j * 64 + lane));
            __builtin_amdgcn_sched_barrier(0);               // issue order = consume order (the loop's counted waits rely on it)
        }
    }
    // activations: fragment (chunk, row tile) of the prompt's packed [k / 32][mtiles][lane][8]; a ragged last block re-reads the last row tile (never stored)
    const T* X = reinterpret_cast<const T*>(a.X);
    u4 xf[XP_MT][XP_CPW];
#pragma unroll
    for (int c = 0; c < XP_CPW; ++c)                         // chunk-major: the first trip consumes chunk by chunk, so its waits shrink with the queue
#pragma unroll
        for (int mt = 0; mt < XP_MT; ++mt) {
            const int mtg = min(XP_MT * mb + mt, a.mtiles - 1);
            xf[mt][c] = ldg16_u(X + ((size_t)(((cw0 + c) * a.mtiles + mtg) * 64) << 3), (unsigned)lane * 16u);     // uniform base + lane offset
        }
    __builtin_amdgcn_sched_barrier(0);

    const int rows_pad = a.mtiles * 16;
    float* sl = slab + (size_t)kg * rows_pad * a.N;
    const int e_q = threadIdx.x >> 8, e_idx = threadIdx.x & 255, e_ml = e_idx >> 4, e_nl = e_idx & 15;

    // every trip refills the ring for the next one, the last trip from a clamped tile (32 KiB per workgroup fetched in vain, from L2): with a third, refill-free
    // copy of the body for the last trip (as in xstat32_k) hipcc spilled 13 fragments around the loop
    auto trip = [&](int it) {
        constexpr bool PF = true;
        const int t0 = (wid + it * G) * XP_TPI;
        // D[n_local = 4 g + reg][m_local = r] -> red[buf][q][wave][mt][m_local * 16 + n_local], tile by tile (one tile's 16 accumulator registers live at a time)
        float* rb = red + (size_t)(it & 1) * (XP_TPI * XP_WAVES * XP_MT * 256);
#pragma unroll
        for (int q = 0; q < XP_TPI; ++q) {
            v4f acc[XP_MT];
#pragma unroll
            for (int mt = 0; mt < XP_MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
            const u4* wn = tile_ptr(t0 + G * XP_TPI + q);
#pragma unroll
            for (int j = 0; j < XP_CPW; ++j) {
                const u4 wv = ring[q * XP_CPW + j];
#pragma unroll
                for (int mt = 0; mt < XP_MT; ++mt) acc[mt] = mfma16(as_vec8<T>(wv), as_vec8<T>(xf[mt][j]), acc[mt]);
                if (PF) ring[q * XP_CPW + j] = ldg16_nt(wn + (unsigned)(j * 64 + lane));
                __builtin_amdgcn_sched_barrier(0);           // consume-j / refill-j order: the waits stay counted
            }
#pragma unroll
            for (int mt = 0; mt < XP_MT; ++mt)
                *reinterpret_cast<float4*>(&rb[((q * XP_WAVES + wa) * XP_MT + mt) * 256 + r * 16 + g * 4]) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
        }
        __syncthreads();
        const int t_o = t0 + e_q, n = t_o * 16 + e_nl;
#pragma unroll
        for (int mt = 0; mt < XP_MT; ++mt) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < XP_WAVES; ++i) v += rb[((e_q * XP_WAVES + i) * XP_MT + mt) * 256 + e_idx];
            const int row = (XP_MT * mb + mt) * 16 + e_ml;
            if (t_o < ntiles && n < a.N && XP_MT * mb + mt < a.mtiles && row < a.M) sl[(size_t)row * a.N + n] = v;
        }
    };
    // the first trip is peeled: its waits cover the activation loads (newest in the queue), the loop's stay counted on the ring alone
    trip(0);
    for (int it = 1; it < nit; ++it) trip(it);
}

// gate/up: out = swiglu(T(s0 + s1)) per (row, pair of gate / up columns) -> the prompt's fragment-packed order for down_proj (xpacked 3). The weight tiles
// interleave 8 gate rows and the 8 matching up rows (weights.py), so columns 16 t .. + 8 are gate 8 t .., 16 t + 8 .. + 8 the matching up. One thread = one
// 16-byte packed piece (8 outputs of one row); rows >= M of the last row tile are zero-filled.
template <typename T>
__global__ __launch_bounds__(256) void swiglu_slab_k(const float* __restrict__ slab, T* __restrict__ out, int M, int mtiles, int N) {
    typedef typename Vec8<T>::type V8;
    const int nt = N >> 4, rows_pad = mtiles * 16;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows_pad * nt) return;
    const int t = (int)(i / rows_pad), m = (int)(i - (long)t * rows_pad);        // consecutive threads: consecutive rows of one tile (16-byte pieces are contiguous per 16 rows)
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fromf<T>(0.f);
    if (m < M) {
        const float* s0 = slab + (size_t)m * N + t * 16;
        const float* s1 = s0 + (size_t)rows_pad * N;
        float gv[8], uv[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 a0 = *reinterpret_cast<const float4*>(s0 + 4 * q), a1 = *reinterpret_cast<const float4*>(s1 + 4 * q);
            const float4 b0 = *reinterpret_cast<const float4*>(s0 + 8 + 4 * q), b1 = *reinterpret_cast<const float4*>(s1 + 8 + 4 * q);
            gv[4 * q] = a0.x + a1.x; gv[4 * q + 1] = a0.y + a1.y; gv[4 * q + 2] = a0.z + a1.z; gv[4 * q + 3] = a0.w + a1.w;
            uv[4 * q] = b0.x + b1.x; uv[4 * q + 1] = b0.y + b1.y; uv[4 * q + 2] = b0.z + b1.z; uv[4 * q + 3] = b0.w + b1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fromf<T>(swiglu<T>(gv[j], uv[j]));
    }
    stg16(out + ((((size_t)(t >> 2) * mtiles + (m >> 4)) * 64 + (t & 3) * 16 + (m & 15)) << 3), as_u4<T>(o));
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
// X fragment-packed (xpacked 3) over a.mtiles row tiles, 65 <= M <= 192, K = 4096, model-dtype weights; slab needs 2 x 16 mtiles x N floats
bool xprompt64_supported(const GemmArgs& a) {
    return a.xpacked == 3 && a.mtiles >= 5 && a.mtiles <= 12 && a.M <= a.mtiles * 16 && a.M > (a.mtiles - 1) * 16 && a.K == XP_K && a.W && !a.W8 && !a.norm_w && !a.bias &&
           a.N % 16 == 0 && (a.N + 15) / 16 >= 128;
}

void launch_xprompt64(int dtype, const GemmArgs& a, float* slab, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, {
        static DevOnce attr;
        if (attr.first()) (void)hipFuncSetAttribute((const void*)xprompt64_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)XP_SMEM);
        hipLaunchKernelGGL((xprompt64_k<T>), dim3(256), dim3(XP_THREADS), XP_SMEM, s, a, slab);
    });
}

void launch_swiglu_slab(int dtype, const float* slab, void* out, int M, int mtiles, int N, hipStream_t s) {
    const long total = (long)mtiles * 16 * (N / 16);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((swiglu_slab_k<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slab, (T*)out, M, mtiles, N));
}

}  // namespace rdx

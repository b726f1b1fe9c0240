// Fragment-packed convolution / GEMM for the image encoder (round 4): BOTH operands arrive in MFMA fragment order and go straight from
// global memory (L2) into registers -- no LDS staging, no barriers, no LDS-DMA issue cost, every wave independent.
//
// Why: the tile GEMMs of rounds 1-3 (gemm_dma_k: 128 x 128 x 64 steps through LDS) spend 0.7-1.0 us per k-step whatever the grid --
// each wave issues 8 LDS-DMA pieces (60-185 cycles of issue each) and 16 ds_read_b128 per 32 MFMAs, behind two barriers -- and a
// convolution with a few hundred row tiles cannot amortise that: the deep trunk stages ran at 370-590 TFLOP/s at batch 32 and at
// 30-60 TFLOP/s at batch 1 (profiles/r03_encoder_shapes.md). The weights were always stored in fragment order
// ([n / 16][k / 32][lane (g, r)][8], one coalesced KiB per MFMA A operand); this kernel family gives the ACTIVATIONS the same
// treatment, as the decode path did in round 1:
//
//   packed activation tensor P[c / 32][m / 16][lane (g, r)][8] = X[m = 16 mt + r][c = 32 kc + 8 g .. + 8],  m = (b H + h) W + w (NHWC pixel)
//
// so that a wave's MFMA B operand (16 pixels x 32 channels) is ONE contiguous KiB too. A wave owns MTW x NTW output tiles of 16 pixels x
// 16 channels and walks K = taps x Cin in 32-deep chunks: NTW weight fragments + MTW activation fragments per chunk (16-byte lane loads,
// the next chunk's loads in flight under this chunk's MFMAs), MTW x NTW MFMAs.
//   * 1 x 1 stride 1: the activation fragments are whole tiles of P (pure streaming);
//   * 3 x 3 (pad 1) and stride 2: implicit GEMM by ADDRESS -- lane (g, r) of m-tile i fetches the 16 bytes of input pixel
//     (b, s oh + dy, s ow + dx), i.e. slot (g, m_in % 16) of tile m_in / 16 of the same channel chunk: a quarter wave reads 16 consecutive
//     slots of at most two neighbouring tiles (stride 1) -- contiguous 256 bytes, not a 16-row gather. Taps in the padding read a zero
//     line (stride 0 across chunks);
//   * epilogue without LDS: an accumulator lane holds 4 channels of one pixel; two column tiles (n even / odd = the two halves of a
//     32-channel chunk) are exchanged between lane rows with v_permlane16_swap so that every lane ends with 8 consecutive channels = one
//     16-byte slot of the PACKED output -- a wave stores a complete, contiguous KiB fragment block per (chunk, m-tile); the residual is the
//     same slot of the residual tensor. Rounding points as the tiled kernels: T(acc + bias) [relu], then T(relu(resid + that)).
//   * small grids (one image's deep stages, the Q-Former's 32 .. 1024-row GEMMs): the 4 or 8 waves of a workgroup share ONE output tile and split K (KSPLIT);
//     partials meet in LDS in wave order (deterministic) -- a wave's serial walk over K, not the byte count, is what such a launch waits for.
// What bounds it: the CU's vector-memory path (64 B/clk): (MTW + NTW) KiB per MTW x NTW MFMAs -- 4 x 4 tiles cap at 50 % of the MFMA peak (23-25 % reached
// at batch 32, PMC MfmaUtil 16-21 % over the 3 x 3 shapes: profiles/r04_pconv_shapes.md, r04_pmc_encoder_b32.md); the memory-bound stages (layer1 / layer2)
// stream at 4.1-4.5 TB/s. Also in this file: the row-major <-> packed converters of the test hook, the packed LayerNorm and query broadcast of the Q-Former.
#include <algorithm>
#include <stdlib.h>

#include <stdio.h>
#include <stdlib.h>
#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

// ---- layout conversion: row-major [M][C] <-> packed [C / 32][mtiles][64][8] ---------------------------------------------------------
// one thread per 16-byte slot; rows beyond M are written as zeros
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_k(const T* __restrict__ X, int ldx, u4* __restrict__ P, int M, int C, int mtiles) {
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)(C / 32) * mtiles * 64;
    if (slot >= total) return;
    const int lane = slot & 63, r = lane & 15, g = lane >> 4;
    const size_t t = slot >> 6;
    const int mt = t % mtiles, kc = t / mtiles, m = mt * 16 + r;
    u4 v = (u4){0u, 0u, 0u, 0u};
    if (m < M) v = ldg16(X + (size_t)m * ldx + kc * 32 + g * 8);
    P[slot] = v;
}
template <typename T>
__global__ __launch_bounds__(256) void unpack_rows_k(const u4* __restrict__ P, T* __restrict__ X, int ldx, int M, int C, int mtiles) {
    // thread -> (row m, 8-channel group): coalesced on the row-major side
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)M * (C / 8);
    if (i >= total) return;
    const int c8 = i % (C / 8), m = i / (C / 8);
    const int kc = c8 >> 2, g = c8 & 3, mt = m >> 4, r = m & 15;
    stg16(X + (size_t)m * ldx + c8 * 8, P[((size_t)kc * mtiles + mt) * 64 + g * 16 + r]);
}

void launch_pack_rows(int dtype, const void* X, int ldx, void* P, int M, int C, hipStream_t s) {
    const int mtiles = (M + 15) / 16;
    const size_t total = (size_t)(C / 32) * mtiles * 64;
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((pack_rows_k<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)X, ldx, (u4*)P, M, C, mtiles));
}
void launch_unpack_rows(int dtype, const void* P, void* X, int ldx, int M, int C, hipStream_t s) {
    const int mtiles = (M + 15) / 16;
    const size_t total = (size_t)M * (C / 8);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((unpack_rows_k<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const u4*)P, (T*)X, ldx, M, C, mtiles));
}

// ---- LayerNorm over the channels of a packed tensor (Q-Former post-LN; fp32 statistics, two-pass variance like layernorm_k) ------------
// one 4-wave workgroup per 16-row tile (a first version with ONE wave per tile took 13.6 us at 32 rows: 24 dependent fragment loads and 190 values per
// lane in a single wave). out_f32 (nullable) = the rows in fp32 row-major [M][H] (the Q-Former's last_hidden_state).
template <typename T, int KCW>
__global__ __launch_bounds__(256) void layernorm_packed_k(const u4* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          u4* __restrict__ out, float* __restrict__ out_f32, int H, int mtiles, int M, float eps) {
    // one workgroup (4 waves) per 16-row tile; wave w holds chunks w, w + 4, ... (<= KCW of them): lane (g, r) = row r's channels 32 kc + 8 g .. + 8.
    // Row statistics: sum over a lane's values, over the four lane rows g (v_permlane16/32_swap), over the waves (LDS, fixed order).
    typedef typename Vec8<T>::type V8;
    __shared__ float red[2][4][16];
    const int mt = blockIdx.x, lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
    const int KC = H >> 5;
    u4 v[KCW];
    float gm[KCW][8], bt[KCW][8];
#pragma unroll
    for (int q = 0; q < KCW; ++q) {
        const int kc = w + 4 * q;
        if (kc < KC) {
            v[q] = ldg16(x + ((size_t)kc * mtiles + mt) * 64 + lane);
            const int c0 = kc * 32 + g * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
            gm[q][0] = g0.x; gm[q][1] = g0.y; gm[q][2] = g0.z; gm[q][3] = g0.w; gm[q][4] = g1.x; gm[q][5] = g1.y; gm[q][6] = g1.z; gm[q][7] = g1.w;
            bt[q][0] = b0.x; bt[q][1] = b0.y; bt[q][2] = b0.z; bt[q][3] = b0.w; bt[q][4] = b1.x; bt[q][5] = b1.y; bt[q][6] = b1.z; bt[q][7] = b1.w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < KCW; ++q)
        if (w + 4 * q < KC) {
            const V8 e = as_vec8<T>(v[q]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += tof<T>(e[j]);
        }
    s = xor32_sum(xor16_sum(s));
    if (g == 0) red[0][w][r] = s;
    __syncthreads();
    const float mean = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) / (float)H;
    float qv = 0.f;
#pragma unroll
    for (int q = 0; q < KCW; ++q)
        if (w + 4 * q < KC) {
            const V8 e = as_vec8<T>(v[q]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = tof<T>(e[j]) - mean; qv += d * d; }
        }
    qv = xor32_sum(xor16_sum(qv));
    if (g == 0) red[1][w][r] = qv;
    __syncthreads();
    const float rstd = rsqrtf(((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) / (float)H + eps);
    const int m = mt * 16 + r;
#pragma unroll
    for (int q = 0; q < KCW; ++q) {
        const int kc = w + 4 * q;
        if (kc < KC) {
            const V8 e = as_vec8<T>(v[q]);
            V8 o;
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { y[j] = (tof<T>(e[j]) - mean) * rstd * gm[q][j] + bt[q][j]; o[j] = fromf<T>(y[j]); }
            if (out) out[((size_t)kc * mtiles + mt) * 64 + lane] = as_u4<T>(o);
            if (out_f32 && m < M) {
                float4* d = reinterpret_cast<float4*>(out_f32 + (size_t)m * H + kc * 32 + g * 8);
                d[0] = make_float4(y[0], y[1], y[2], y[3]);
                d[1] = make_float4(y[4], y[5], y[6], y[7]);
            }
        }
    }
}

bool layernorm_packed_supported(int H) { return H % 32 == 0 && H <= 1024; }
void launch_layernorm_packed(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* out_f32, int M, int H, float eps,
                             hipStream_t s) {
    const int mtiles = (M + 15) / 16;
    RDX_DISPATCH_T(dtype, T, {
        if (H <= 256) hipLaunchKernelGGL((layernorm_packed_k<T, 2>), dim3(mtiles), dim3(256), 0, s, (const u4*)x, gamma, beta, (u4*)out, out_f32, H, mtiles, M, eps);
        else if (H <= 768) hipLaunchKernelGGL((layernorm_packed_k<T, 6>), dim3(mtiles), dim3(256), 0, s, (const u4*)x, gamma, beta, (u4*)out, out_f32, H, mtiles, M, eps);
        else hipLaunchKernelGGL((layernorm_packed_k<T, 8>), dim3(mtiles), dim3(256), 0, s, (const u4*)x, gamma, beta, (u4*)out, out_f32, H, mtiles, M, eps);
    });
}

// the `rows` x H row-major block `src` (LayerNorm(query_tokens), rows % 16 == 0) repeated for B images, packed: tile mt holds rows (mt % (rows / 16)) * 16 ..
template <typename T>
__global__ __launch_bounds__(256) void broadcast_packed_k(const T* __restrict__ src, u4* __restrict__ dst, int rows, int H, int mtiles) {
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)(H / 32) * mtiles * 64;
    if (slot >= total) return;
    const int lane = slot & 63, r = lane & 15, g = lane >> 4;
    const size_t t = slot >> 6;
    const int mt = t % mtiles, kc = t / mtiles, row = (mt % (rows / 16)) * 16 + r;
    dst[slot] = ldg16(src + (size_t)row * H + kc * 32 + g * 8);
}
void launch_broadcast_packed(int dtype, const void* src, void* dst, int rows, int H, int B, hipStream_t s) {
    const int mtiles = B * rows / 16;
    const size_t total = (size_t)(H / 32) * mtiles * 64;
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((broadcast_packed_k<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)src, (u4*)dst, rows, H, mtiles));
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------
// TAPS 1 | 9 (3 x 3, pad 1); STRIDE 1 | 2; ROWOUT: the output is written row-major [M][ldo] instead of packed (the trunk's last block)
// KSPLIT: the workgroup's waves (4 or 8 = blockDim / 64) share ONE output tile and split its K chunks between them; partials meet in LDS in a
// fixed order (deterministic). For grids that cannot give every SIMD a wave (the Q-Former's 1024-row GEMMs: 192-576 tiles; the deep convolutions of a
// single image) this shortens a wave's serial walk over K -- 24-144 dependent L2 round trips -- by the split factor at the same total traffic.
template <typename T, int EPI, int MTW, int NTW, int TAPS, int STRIDE, bool ROWOUT, bool KSPLIT>
__global__ __launch_bounds__(KSPLIT ? 512 : 256) void pconv_k(PConvArgs a) {
    constexpr int NS = MTW * NTW >= 16 ? 2 : (MTW * NTW >= 8 ? 3 : 4);       // register-ring depth
    typedef typename Vec8<T>::type V8;
    static_assert(NTW % 2 == 0, "column tiles are paired in the epilogue");
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int n_groups = (a.N >> 4) / NTW;
    const int gw = KSPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + w;
    const int mg = gw / n_groups, ng = gw - mg * n_groups;
    const int m_groups = (a.mt_out + MTW - 1) / MTW;
    if (mg >= m_groups) return;                                    // (never with KSPLIT: the grid is exact, every wave reaches the barrier)
    const int clog = a.clog, cmask = TAPS > 1 ? (1 << clog) - 1 : 0x7fffffff;       // 3 x 3: Cin / 32 = 1 << clog chunks per tap (a power of two)
    const int KC = TAPS > 1 ? TAPS << clog : a.Cin >> 5;           // 32-deep chunks of K
    // this wave's chunk range
    int k_lo = 0, k_hi = KC;
    if (KSPLIT) {
        const int nw = blockDim.x >> 6, per = (KC + nw - 1) / nw;
        k_lo = min(w * per, KC); k_hi = min(k_lo + per, KC);
    }
    const u4* Xp = reinterpret_cast<const u4*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W) + (size_t)(ng * NTW) * KC * 64 + lane;
    const u4* zero = reinterpret_cast<const u4*>(a.zero16);
    const unsigned slab = (unsigned)a.mt_in * 64u;                 // 16-byte slots per channel chunk of the input

    // pixel of this lane in each of the wave's m-tiles
    int pb[MTW], ph[MTW], pw[MTW];
    constexpr bool GATHER = TAPS > 1 || STRIDE > 1;
    if (GATHER) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            int m = (mg * MTW + i) * 16 + r;
            m = min(m, a.M - 1);
            const int t = m / a.Wout;
            pw[i] = (m - t * a.Wout) * STRIDE;
            pb[i] = t / a.Hout;
            ph[i] = (t - pb[i] * a.Hout) * STRIDE;
        }
    }
    const u4* xptr[MTW];
    unsigned xstr[MTW];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            if (!GATHER) {
                const int mt = min(mg * MTW + i, a.mt_in - 1);
                xptr[i] = Xp + (size_t)mt * 64 + lane;
                xstr[i] = slab;
            } else {
                const int dy = TAPS > 1 ? tap / 3 - 1 : 0, dx = TAPS > 1 ? tap % 3 - 1 : 0;
                const int ih = ph[i] + dy, iw = pw[i] + dx;
                const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
                const int m_in = (pb[i] * a.Hin + ih) * a.Win + iw;
                const u4* p = Xp + (size_t)(m_in >> 4) * 64 + g * 16 + (m_in & 15);
                xptr[i] = ok ? p : zero;
                xstr[i] = ok ? slab : 0u;
            }
        }
    };
    auto load = [&](int kk, u4 (&wf)[NTW], u4 (&xf)[MTW]) {
        const unsigned ckc = (unsigned)(kk & cmask);
#pragma unroll
        for (int j = 0; j < NTW; ++j) wf[j] = ldg16(Wp + ((size_t)j * KC + kk) * 64);
#pragma unroll
        for (int i = 0; i < MTW; ++i) xf[i] = ldg16(xptr[i] + (size_t)ckc * xstr[i]);
    };
    v4f acc[NTW][MTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < MTW; ++i) acc[j][i] = (v4f){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const u4 (&wf)[NTW], const u4 (&xf)[MTW]) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int i = 0; i < MTW; ++i) acc[j][i] = mfma16(as_vec8<T>(wf[j]), as_vec8<T>(xf[i]), acc[j][i]);
    };
    // K loop over a register ring of NS stages: the loads of chunk kk + NS are issued right behind the MFMAs that consumed chunk kk's
    // stage, so NS - 1 chunks are always in flight (small tiles are latency-bound per chunk: a deeper ring, not a wider tile, is what
    // shortens a wave's serial K walk)
    u4 wr[NS][NTW], xr[NS][MTW];
    int tap_set = TAPS > 1 ? k_lo >> clog : 0;
    set_tap(tap_set);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int kn = k_lo + s_;
        if (kn < k_hi) {
            if (TAPS > 1 && (kn >> clog) != tap_set) { tap_set = kn >> clog; set_tap(tap_set); }
            load(kn, wr[s_], xr[s_]);
        }
    }
    for (int kk = k_lo; kk < k_hi; kk += NS) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (kk + s_ < k_hi) {
                mma(wr[s_], xr[s_]);
                const int kn = kk + s_ + NS;
                if (kn < k_hi) {
                    if (TAPS > 1 && (kn >> clog) != tap_set) { tap_set = kn >> clog; set_tap(tap_set); }
                    load(kn, wr[s_], xr[s_]);
                }
            }
        }
    }
    if (KSPLIT) {
        // partial tiles -> LDS [wave][j][i][lane] (16 bytes per lane); the wave that owns m-tile i (i % waves) adds the partials of its tiles in
        // wave order and runs their epilogue
        extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
        v4f* part = reinterpret_cast<v4f*>(psm);
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int i = 0; i < MTW; ++i) part[((w * NTW + j) * MTW + i) * 64 + lane] = acc[j][i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            if ((i % nw) != w) continue;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                v4f t = part[((0 * NTW + j) * MTW + i) * 64 + lane];
                for (int ww = 1; ww < nw; ++ww) {
                    const v4f p = part[((ww * NTW + j) * MTW + i) * 64 + lane];
                    t[0] += p[0]; t[1] += p[1]; t[2] += p[2]; t[3] += p[3];
                }
                acc[j][i] = t;
            }
        }
    }

    // ---- epilogue: pairs of column tiles -> one 16-byte slot per lane of the packed output (or a 16-byte row segment) ---------------
    const bool odd_row = g & 1;
#pragma unroll
    for (int jp = 0; jp < NTW; jp += 2) {
        const int nt0 = ng * NTW + jp;                             // even column tile of the pair; output chunk kc_out = nt0 / 2
        float4 be = make_float4(0.f, 0.f, 0.f, 0.f), bo = be;
        if (a.bias) {
            be = *reinterpret_cast<const float4*>(a.bias + nt0 * 16 + g * 4);
            bo = *reinterpret_cast<const float4*>(a.bias + nt0 * 16 + 16 + g * 4);
        }
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mt = mg * MTW + i;
            if (mt >= a.mt_out) continue;
            if (KSPLIT && (i % (int)(blockDim.x >> 6)) != w) continue;      // the owner of m-tile i holds the reduced tile
            float ve[4], vo[4];
            ve[0] = acc[jp][i][0] + be.x; ve[1] = acc[jp][i][1] + be.y; ve[2] = acc[jp][i][2] + be.z; ve[3] = acc[jp][i][3] + be.w;
            vo[0] = acc[jp + 1][i][0] + bo.x; vo[1] = acc[jp + 1][i][1] + bo.y; vo[2] = acc[jp + 1][i][2] + bo.z; vo[3] = acc[jp + 1][i][3] + bo.w;
            if (EPI == EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ve[e] = fmaxf(ve[e], 0.f); vo[e] = fmaxf(vo[e], 0.f); }
            }
            if (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ve[e] = gelu_erf(ve[e]); vo[e] = gelu_erf(vo[e]); }
            }
            // round to T and pack pairs: e -> 2 dwords, o -> 2 dwords
            unsigned e0 = bits16<T>(fromf<T>(ve[0])) | ((unsigned)bits16<T>(fromf<T>(ve[1])) << 16);
            unsigned e1 = bits16<T>(fromf<T>(ve[2])) | ((unsigned)bits16<T>(fromf<T>(ve[3])) << 16);
            unsigned o0 = bits16<T>(fromf<T>(vo[0])) | ((unsigned)bits16<T>(fromf<T>(vo[1])) << 16);
            unsigned o1 = bits16<T>(fromf<T>(vo[2])) | ((unsigned)bits16<T>(fromf<T>(vo[3])) << 16);
            // v_permlane16_swap(e, o): odd lane rows of e <-> even lane rows of o. Afterwards an even row g' holds the even tile's channels
            // 4 g' .. 4 g' + 7 (its own e, its upper neighbour's e in `o`), an odd row the odd tile's channels 4 (g' - 1) .. + 7
            // (its lower neighbour's o in `e`, its own o).
            const auto s0 = __builtin_amdgcn_permlane16_swap(e0, o0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(e1, o1, false, false);
            u4 v = (u4){(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
            // slot of this lane inside the 32-channel chunk: even rows -> g' / 2 (even tile), odd rows -> 2 + g' / 2 (odd tile)
            const int gs = (odd_row ? 2 : 0) + (g >> 1);
            const int m = mt * 16 + r;
            if (EPI == EPI_RESID_RELU || EPI == EPI_RESID) {
                const u4 rv = ldg16(reinterpret_cast<const u4*>(a.resid) + ((size_t)(nt0 >> 1) * a.mt_out + mt) * 64 + gs * 16 + r);
                const V8 cv = as_vec8<T>(v), rr = as_vec8<T>(rv);
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = tof<T>(rr[e]) + tof<T>(cv[e]);
                    if (EPI == EPI_RESID_RELU) s = fmaxf(s, 0.f);
                    o[e] = fromf<T>(s);
                }
                v = as_u4<T>(o);
            }
            if (ROWOUT) {
                if (m < a.M) stg16(reinterpret_cast<T*>(a.out) + (size_t)m * a.ldo + nt0 * 16 + gs * 8, v);
            } else {
                if (m >= a.M) v = (u4){0u, 0u, 0u, 0u};             // pad rows of the last tile stay zero
                stg16(reinterpret_cast<u4*>(a.out) + ((size_t)(nt0 >> 1) * a.mt_out + mt) * 64 + gs * 16 + r, v);
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

bool pconv_supported(const PConvArgs& a, int taps, int stride, int epi) {
    if (!(taps == 1 || taps == 9) || !(stride == 1 || stride == 2)) return false;
    if (a.Cin < 32 || a.Cin % 32 || a.N % 32) return false;                       // whole 32-deep chunks; column tiles in pairs
    if (taps == 9 && (a.Cin & (a.Cin - 1))) return false;                         // 3 x 3: a power-of-two channel count (ResNet-50: 64 .. 512)
    if (epi == EPI_GELU || epi == EPI_RESID) return taps == 1 && stride == 1 && (epi == EPI_GELU || a.resid);      // the plain-GEMM epilogues
    if (!(epi == EPI_NONE || epi == EPI_RELU || epi == EPI_RESID_RELU)) return false;
    if (epi == EPI_RESID_RELU && !a.resid) return false;
    return true;
}

// tile shape: the largest register tile that still gives every SIMD a wave (256 CUs x 4 SIMDs = 1024). When no tile does and K is long enough, the
// waves of a workgroup split K (KSPLIT, 4 or 8 waves per tile) -- the grid grows by that factor at the same total traffic.
// Measured at batch 32 / batch 1 over the 23 trunk shapes (tools/pconv_check.py, profiles/r04_pconv_shapes.md): within 10 % of the best tile per shape.
void pconv_pick(const PConvArgs& a, int taps, int* mtw, int* ntw, int* ks) {
    const int nt = a.N / 16, KC = taps * (a.Cin / 32);
    *ks = 1;
    if (a.tile_m > 0 && (a.tile_n == 2 || a.tile_n == 4) && nt % a.tile_n == 0) {      // forced tile (the debug hook's RDX_PCONV_TILE)
        *mtw = a.tile_m; *ntw = a.tile_n;
        if (a.tile_k == 4 || a.tile_k == 8) *ks = (a.tile_k == 8 && *mtw * *ntw > 8) ? 4 : a.tile_k;
        return;
    }
    auto tiles = [&](int mm, int nn) { return (long)((a.mt_out + mm - 1) / mm) * (nt / nn); };
    const long t44 = nt % 4 == 0 ? tiles(4, 4) : 0;
    // (1) big grids (the trunk at batch): 64 x 64 tiles, one wave each, no split -- a K-split workgroup's LDS partials limit residency and a
    //     784-tile layer4 convolution then runs in 1.5 rounds (139 vs 69 us)
    if (t44 >= 1024 || (t44 >= 700 && a.M >= 4096)) { *mtw = 4; *ntw = 4; return; }
    const bool allow_ks = !a.no_ksplit;
    // (2) grids that cannot give every SIMD a wave and a K of >= 16 chunks: the workgroup's waves split K (Q-Former GEMMs at 1024 rows: 76 -> 45 us
    //     per layer; layer3 / layer4 of a single image: 11-19 -> 6-7 us)
    if (allow_ks && KC >= 16) {
        if (t44 * 4 >= 1024 && t44 <= 640) { *mtw = 4; *ntw = 4; *ks = 4; return; }
        if (tiles(4, 2) * 4 >= 1024) { *mtw = 4; *ntw = 2; *ks = 4; return; }
        const int k = KC >= 64 ? 8 : 4;
        if (tiles(2, 2) * k >= 1024) { *mtw = 2; *ntw = 2; *ks = k; return; }
        *mtw = 1; *ntw = 2; *ks = k;
        return;
    }
    // (3) short K: the largest tile that still gives every SIMD a wave, else the smallest tile
    static const int cand[5][2] = {{4, 4}, {4, 2}, {2, 4}, {2, 2}, {1, 2}};
    for (int i = 0; i < 5; ++i) {
        if (nt % cand[i][1]) continue;
        *mtw = cand[i][0]; *ntw = cand[i][1];
        if (tiles(*mtw, *ntw) >= 1024) return;
    }
}

template <typename T, int EPI, int MTW, int NTW, int TAPS, int STRIDE>
static void launch_pc5(const PConvArgs& a, bool rowout, int ks, hipStream_t s) {
    const int n_groups = (a.N / 16) / NTW, m_groups = (a.mt_out + MTW - 1) / MTW;
    const long waves = (long)n_groups * m_groups;
    if (ks > 1) {
        const dim3 grid((unsigned)waves), block(64 * ks);
        const size_t smem = (size_t)ks * MTW * NTW * 1024;
        if (rowout) hipLaunchKernelGGL((pconv_k<T, EPI, MTW, NTW, TAPS, STRIDE, true, true>), grid, block, smem, s, a);
        else hipLaunchKernelGGL((pconv_k<T, EPI, MTW, NTW, TAPS, STRIDE, false, true>), grid, block, smem, s, a);
        return;
    }
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (rowout) hipLaunchKernelGGL((pconv_k<T, EPI, MTW, NTW, TAPS, STRIDE, true, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((pconv_k<T, EPI, MTW, NTW, TAPS, STRIDE, false, false>), grid, block, 0, s, a);
}
template <typename T, int EPI, int MTW, int NTW>
static void launch_pc3(const PConvArgs& a, int taps, int stride, bool rowout, int ks, hipStream_t s) {
    if (taps == 1 && stride == 1) launch_pc5<T, EPI, MTW, NTW, 1, 1>(a, rowout, ks, s);
    else if (EPI == EPI_GELU || EPI == EPI_RESID) return;          // plain-GEMM epilogues (pconv_supported)
    else if (taps == 1) launch_pc5<T, (EPI == EPI_GELU || EPI == EPI_RESID) ? EPI_NONE : EPI, MTW, NTW, 1, 2>(a, rowout, ks, s);
    else if (stride == 1) launch_pc5<T, (EPI == EPI_GELU || EPI == EPI_RESID) ? EPI_NONE : EPI, MTW, NTW, 9, 1>(a, rowout, ks, s);
    else launch_pc5<T, (EPI == EPI_GELU || EPI == EPI_RESID) ? EPI_NONE : EPI, MTW, NTW, 9, 2>(a, rowout, ks, s);
}
template <typename T, int EPI>
static void launch_pc1(const PConvArgs& a, int taps, int stride, bool rowout, int mtw, int ntw, int ks, hipStream_t s) {
    if (ntw == 4) {
        if (mtw >= 4) launch_pc3<T, EPI, 4, 4>(a, taps, stride, rowout, ks, s);
        else launch_pc3<T, EPI, 2, 4>(a, taps, stride, rowout, ks, s);
    } else {
        if (mtw >= 4) launch_pc3<T, EPI, 4, 2>(a, taps, stride, rowout, ks, s);
        else if (mtw == 2) launch_pc3<T, EPI, 2, 2>(a, taps, stride, rowout, ks, s);
        else launch_pc3<T, EPI, 1, 2>(a, taps, stride, rowout, ks, s);
    }
}

bool launch_pconv(int dtype, PConvArgs a, int taps, int stride, int epi, bool rowout, hipStream_t s) {
    // a 1 x 1 stride-1 convolution reads input tile mt for output tile mt (no gather): the two tensors must have the same tiling. Nothing is
    // launched otherwise and the caller reports through the ABI's error path (rdx_ctx::unsupported -> -8): the library never aborts its host process
    if (taps == 1 && stride == 1 && a.mt_in != a.mt_out) return false;
    a.clog = ilog2(a.Cin / 32);
    int mtw = 1, ntw = 2, ks = 1;
    pconv_pick(a, taps, &mtw, &ntw, &ks);
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_RELU) launch_pc1<T, EPI_RELU>(a, taps, stride, rowout, mtw, ntw, ks, s);
        else if (epi == EPI_RESID_RELU) launch_pc1<T, EPI_RESID_RELU>(a, taps, stride, rowout, mtw, ntw, ks, s);
        else if (epi == EPI_GELU) launch_pc1<T, EPI_GELU>(a, taps, stride, rowout, mtw, ntw, ks, s);
        else if (epi == EPI_RESID) launch_pc1<T, EPI_RESID>(a, taps, stride, rowout, mtw, ntw, ks, s);
        else launch_pc1<T, EPI_NONE>(a, taps, stride, rowout, mtw, ntw, ks, s);
    });
    return true;
}

}  // namespace rdx

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
// What bounds it: the CU's vector-memory path (64 B/clk): (MTW + NTW) KiB per MTW x NTW MFMAs -- 4 x 4 tiles cap at 50 % of the MFMA peak, 8 x 4
// at 67 %; the memory-bound stages (layer1 / layer2) stream at the HBM rate.
#include <algorithm>
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

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------
// TAPS 1 | 9 (3 x 3, pad 1); STRIDE 1 | 2; ROWOUT: the output is written row-major [M][ldo] instead of packed (the trunk's last block)
template <typename T, int EPI, int MTW, int NTW, int TAPS, int STRIDE, bool ROWOUT>
__global__ __launch_bounds__(256) void pconv_k(PConvArgs a) {
    constexpr int NS = MTW * NTW >= 16 ? 2 : (MTW * NTW >= 8 ? 3 : 4);       // register-ring depth
    typedef typename Vec8<T>::type V8;
    static_assert(NTW % 2 == 0, "column tiles are paired in the epilogue");
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int n_groups = (a.N >> 4) / NTW;
    const int gw = blockIdx.x * 4 + w;
    const int mg = gw / n_groups, ng = gw - mg * n_groups;
    const int m_groups = (a.mt_out + MTW - 1) / MTW;
    if (mg >= m_groups) return;
    const int clog = a.clog, cmask = TAPS > 1 ? (1 << clog) - 1 : 0x7fffffff;       // 3 x 3: Cin / 32 = 1 << clog chunks per tap (a power of two)
    const int KC = TAPS > 1 ? TAPS << clog : a.Cin >> 5;           // 32-deep chunks of K
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
    int tap_set = 0;
    set_tap(0);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ < KC) {
            if (TAPS > 1 && (s_ >> clog) != tap_set) { tap_set = s_ >> clog; set_tap(tap_set); }
            load(s_, wr[s_], xr[s_]);
        }
    }
    for (int kk = 0; kk < KC; kk += NS) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (kk + s_ < KC) {
                mma(wr[s_], xr[s_]);
                const int kn = kk + s_ + NS;
                if (kn < KC) {
                    if (TAPS > 1 && (kn >> clog) != tap_set) { tap_set = kn >> clog; set_tap(tap_set); }
                    load(kn, wr[s_], xr[s_]);
                }
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

// tile shape: the largest register tile that still gives every SIMD a wave (256 CUs x 4 SIMDs); below that the smallest tile (most waves).
// Measured at batch 32 / batch 1 over the 23 trunk shapes (tools/pconv_check.py, profiles/r04_pconv_shapes.md): within 10 % of the best
// tile per shape; 8 x 4 (one wave per SIMD by registers) wins only the stride-2 3 x 3 of layer3 by 5 %.
void pconv_pick(const PConvArgs& a, int* mtw, int* ntw) {
    const int nt = a.N / 16;
    const char* e = getenv("RDX_PCONV_TILE");                     // "MxN" override for experiments (tools/pconv_check.py)
    if (e && e[0] >= '1' && e[0] <= '8' && e[1] == 'x' && (e[2] == '2' || e[2] == '4') && nt % (e[2] - '0') == 0) { *mtw = e[0] - '0'; *ntw = e[2] - '0'; return; }
    static const int cand[5][2] = {{4, 4}, {4, 2}, {2, 4}, {2, 2}, {1, 2}};
    auto waves = [&](int mm, int nn) { return (long)((a.mt_out + mm - 1) / mm) * (nt / nn); };
    for (int i = 0; i < 5; ++i) {
        if (nt % cand[i][1]) continue;
        *mtw = cand[i][0]; *ntw = cand[i][1];
        if (waves(*mtw, *ntw) >= 1024) return;
    }
}

template <typename T, int EPI, int MTW, int NTW, int TAPS, int STRIDE>
static void launch_pc5(const PConvArgs& a, bool rowout, hipStream_t s) {
    const int n_groups = (a.N / 16) / NTW, m_groups = (a.mt_out + MTW - 1) / MTW;
    const long waves = (long)n_groups * m_groups;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (rowout) hipLaunchKernelGGL((pconv_k<T, EPI, MTW, NTW, TAPS, STRIDE, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((pconv_k<T, EPI, MTW, NTW, TAPS, STRIDE, false>), grid, block, 0, s, a);
}
template <typename T, int EPI, int MTW, int NTW>
static void launch_pc3(const PConvArgs& a, int taps, int stride, bool rowout, hipStream_t s) {
    if (taps == 1 && stride == 1) launch_pc5<T, EPI, MTW, NTW, 1, 1>(a, rowout, s);
    else if (EPI == EPI_GELU || EPI == EPI_RESID) return;          // plain-GEMM epilogues (pconv_supported)
    else if (taps == 1) launch_pc5<T, (EPI == EPI_GELU || EPI == EPI_RESID) ? EPI_NONE : EPI, MTW, NTW, 1, 2>(a, rowout, s);
    else if (stride == 1) launch_pc5<T, (EPI == EPI_GELU || EPI == EPI_RESID) ? EPI_NONE : EPI, MTW, NTW, 9, 1>(a, rowout, s);
    else launch_pc5<T, (EPI == EPI_GELU || EPI == EPI_RESID) ? EPI_NONE : EPI, MTW, NTW, 9, 2>(a, rowout, s);
}
template <typename T, int EPI>
static void launch_pc1(const PConvArgs& a, int taps, int stride, bool rowout, int mtw, int ntw, hipStream_t s) {
    if (ntw == 4) {
        if (mtw == 8) launch_pc3<T, EPI, 8, 4>(a, taps, stride, rowout, s);
        else if (mtw == 4) launch_pc3<T, EPI, 4, 4>(a, taps, stride, rowout, s);
        else if (mtw == 2) launch_pc3<T, EPI, 2, 4>(a, taps, stride, rowout, s);
        else launch_pc3<T, EPI, 1, 4>(a, taps, stride, rowout, s);
    } else {
        if (mtw >= 4) launch_pc3<T, EPI, 4, 2>(a, taps, stride, rowout, s);
        else if (mtw == 2) launch_pc3<T, EPI, 2, 2>(a, taps, stride, rowout, s);
        else launch_pc3<T, EPI, 1, 2>(a, taps, stride, rowout, s);
    }
}

void launch_pconv(int dtype, PConvArgs a, int taps, int stride, int epi, bool rowout, hipStream_t s) {
    a.clog = ilog2(a.Cin / 32);
    int mtw = 1, ntw = 2;
    pconv_pick(a, &mtw, &ntw);
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_RELU) launch_pc1<T, EPI_RELU>(a, taps, stride, rowout, mtw, ntw, s);
        else if (epi == EPI_RESID_RELU) launch_pc1<T, EPI_RESID_RELU>(a, taps, stride, rowout, mtw, ntw, s);
        else if (epi == EPI_GELU) launch_pc1<T, EPI_GELU>(a, taps, stride, rowout, mtw, ntw, s);
        else if (epi == EPI_RESID) launch_pc1<T, EPI_RESID>(a, taps, stride, rowout, mtw, ntw, s);
        else launch_pc1<T, EPI_NONE>(a, taps, stride, rowout, mtw, ntw, s);
    });
}

}  // namespace rdx

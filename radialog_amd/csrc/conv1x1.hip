// Weight-stationary streaming 1x1 convolution for the memory-bound stages of the ResNet trunk (layer1 / layer2 at batch):
//   out[M][N] = epilogue(X[M][K] . W[N][K]^T),  K = 64 .. 256, N-slice <= 256 columns per workgroup, M = batch x H x W pixels.
//
// At 112^2 / 56^2 resolution these GEMMs move hundreds of MB of activations for a few GFLOP (K = 64: 2 FLOP per output byte): they
// are HBM-streaming kernels, not MFMA kernels. The tiled GEMM (gemm_dma_k: both operands per tile through LDS, one tile per
// workgroup, a lane's epilogue store = 8 bytes of one row) ran them at 2.3 TB/s. Here the loop nest is turned around:
//   * the workgroup's whole weight slice (<= 64 KiB, already in MFMA A-fragment order) is parked in LDS ONCE; 256 persistent
//     workgroups of 8 waves walk the pixel rows;
//   * a wave owns tiles of 32 rows: their MFMA B fragments (16 rows x 32 k per 16-byte lane load; a K = 64 row is one 128-byte
//     line) come straight from global memory into registers, the next tile's loads are issued before this tile's MFMAs;
//   * the accumulators leave through a per-wave LDS transpose (no barrier: one wave, in order) so that residual loads and output
//     stores are 16 bytes per lane with 8 consecutive lanes on one 128-byte row segment -- full-line HBM transactions;
//   * same rounding points as the tiled kernel: T(acc + bias), then relu(resid + that) in fp32, rounded once.
// 1x1 stride-2 downsample convolutions gather their rows (b, 2 oh, 2 ow) by address.
#include <algorithm>
#include <stdlib.h>

#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

constexpr int CS_WAVES = 8, CS_THREADS = CS_WAVES * 64, CS_ROWB = 144;       // scratch row pitch: 64 cols x 2 B + 16 B pad

template <typename T, int EPI, int KC, int NT, int MT>
__global__ __launch_bounds__(CS_THREADS) void conv1x1_stream_k(GemmArgs a, ConvGeom cg, int tiles) {
    typedef typename Vec8<T>::type V8;
    typedef T T4 __attribute__((ext_vector_type(4)));
    constexpr int ROWS = 16 * MT, CHN = NT < 4 ? NT : 4, NCH = NT / CHN;     // 64-column (or narrower) output chunks
    constexpr int LPR = CHN * 2, RPP = 64 / LPR, PASSES = ROWS / RPP;        // lanes per row, rows per pass of the coalesced side
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wl = reinterpret_cast<u4*>(smem);                                    // [NT][KC][64]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    unsigned char* scr = smem + (size_t)NT * KC * 1024 + (size_t)w * ROWS * CS_ROWB;
    float* bl = reinterpret_cast<float*>(smem + (size_t)NT * KC * 1024 + (size_t)CS_WAVES * ROWS * CS_ROWB);      // [NT * 16] bias slice
    const int n_base = blockIdx.y * NT * 16;                                 // this workgroup's column slice
    {   // park the weight slice: NT x KC blocks of 1 KiB, fragment order kept
        const u4* Wp = reinterpret_cast<const u4*>(a.W) + (size_t)(n_base >> 4) * (a.K >> 5) * 64;
        for (int i = threadIdx.x; i < NT * KC * 64; i += CS_THREADS) wl[i] = ldg16(Wp + i);
        for (int i = threadIdx.x; i < NT * 16; i += CS_THREADS) bl[i] = a.bias ? a.bias[n_base + i] : 0.f;
    }
    __syncthreads();
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* R = reinterpret_cast<const T*>(a.resid);
    T* O = reinterpret_cast<T*>(a.out);
    const int HWo = cg.Hout * cg.Wout;
    auto src_row = [&](int m) -> size_t {                                    // input pixel row of output row m
        m = min(m, a.M - 1);
        if (cg.stride <= 1) return (size_t)m;
        const int b = m / HWo, rem = m - b * HWo, oh = rem / cg.Wout, ow = rem - oh * cg.Wout;
        return ((size_t)b * cg.Hin + (size_t)oh * cg.stride) * cg.Win + (size_t)ow * cg.stride;
    };
    auto load_tile = [&](int t, u4 (&xf)[KC][MT]) {
        const int m0 = t * ROWS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const T* p = X + src_row(m0 + 16 * mt + r) * a.ldx + g * 8;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) xf[kc][mt] = ldg16(p + kc * 32);
        }
    };
    const int wave_id = blockIdx.x * CS_WAVES + w, wave_n = gridDim.x * CS_WAVES;
    u4 xa[KC][MT], xb[KC][MT];
    int t = wave_id;
    if (t < tiles) load_tile(t, xa);
    // the coalesced side of the epilogue: lane -> (row rr of a pass, 16-byte piece pc of the chunk's row segment)
    const int rr = lane / LPR, pc = lane % LPR;
    for (; t < tiles; t += wave_n) {
        const int tn = t + wave_n;
        if (tn < tiles) load_tile(tn, xb);                                   // next tile in flight under this tile's MFMAs + epilogue
        const int m0 = t * ROWS;
        u4 rsd[PASSES];
        auto load_resid = [&](int c) {
            if (EPI != EPI_RESID_RELU && EPI != EPI_RESID) return;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int m = min(m0 + p * RPP + rr, a.M - 1);
                rsd[p] = ldg16(R + (size_t)m * a.ldr + n_base + c * CHN * 16 + pc * 8);
            }
        };
        load_resid(0);
        v4f acc[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const V8 wf = as_vec8<T>(wl[(nt * KC + kc) * 64 + lane]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(wf, as_vec8<T>(xa[kc][mt]), acc[nt][mt]);
            }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // MFMA layout (lane = row r, 4 columns at g) -> scratch [row][64 cols], rounded to T with the bias: the conv output
#pragma unroll
            for (int q = 0; q < CHN; ++q) {
                const int nt = c * CHN + q;
                const float4 b4 = *reinterpret_cast<const float4*>(bl + nt * 16 + g * 4);
                const float bias4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[nt][mt][e] + bias4[e];
                        if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                        o[e] = fromf<T>(v);
                    }
                    *reinterpret_cast<T4*>(scr + (16 * mt + r) * CS_ROWB + q * 32 + g * 8) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave, in-order LDS: the chunk is in the scratch
            u4 rcur[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) rcur[p] = rsd[p];
            if (c + 1 < NCH) load_resid(c + 1);
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = p * RPP + rr, m = m0 + row;
                u4 v = *reinterpret_cast<const u4*>(scr + row * CS_ROWB + pc * 16);
                if (EPI == EPI_RESID_RELU || EPI == EPI_RESID) {
                    const V8 cv = as_vec8<T>(v), rv = as_vec8<T>(rcur[p]);
                    V8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float s = tof<T>(rv[e]) + tof<T>(cv[e]);
                        if (EPI == EPI_RESID_RELU) s = fmaxf(s, 0.f);
                        o[e] = fromf<T>(s);
                    }
                    v = as_u4<T>(o);
                }
                if (m < a.M) stg16(O + (size_t)m * a.ldo + n_base + c * CHN * 16 + pc * 8, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // scratch reads done before the next chunk overwrites it
        }
        if (tn < tiles) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xa[kc][mt] = xb[kc][mt];
        }
    }
}

// shapes served: the 1x1 convolutions of the trunk with K <= 256 at batch (enough 32-row tiles to fill 2048 waves) --
// (K, N-slice) = (64, 64) (64, 256) (256, 64) (256, 128) (128, 256); epilogues NONE / RELU / RESID_RELU
static bool cs_shape(int K, int N, int* kc, int* nt) {
    if (K == 64 && N == 64) { *kc = 2; *nt = 4; return true; }
    if (K == 64 && N % 256 == 0) { *kc = 2; *nt = 16; return true; }
    if (K == 256 && N == 64) { *kc = 8; *nt = 4; return true; }
    if (K == 256 && N % 128 == 0) { *kc = 8; *nt = 8; return true; }
    if (K == 128 && N % 256 == 0) { *kc = 4; *nt = 16; return true; }
    return false;
}

bool conv1x1_stream_supported(const GemmArgs& a, const ConvGeom& cg, int epi) {
    constexpr int min_rows = 8192;                               // below that the grid does not cover the chip: the tile GEMMs take it
    int kc, nt;
    if (min_rows <= 0 || a.M < min_rows || a.ldx % 8 || a.ldo % 8 || a.N % 16) return false;
    if (!(epi == EPI_NONE || epi == EPI_RELU || epi == EPI_RESID_RELU)) return false;
    if (epi == EPI_RESID_RELU && (!a.resid || a.ldr % 8)) return false;
    if (cg.mode == 1 && !(cg.KH == 1 && cg.KW == 1 && cg.pad == 0 && cg.Cin == a.K)) return false;
    return cs_shape(a.K, a.N, &kc, &nt);
}

template <typename T, int EPI, int KC, int NT>
static void launch_cs(const GemmArgs& a, const ConvGeom& cg, hipStream_t s) {
    constexpr int MT = 2;
    const int tiles = (a.M + 16 * MT - 1) / (16 * MT), slices = a.N / (NT * 16);
    const size_t smem = (size_t)NT * KC * 1024 + (size_t)CS_WAVES * 16 * MT * CS_ROWB + (size_t)NT * 16 * sizeof(float);
    int gx = std::max(1, 256 / slices);
    gx = std::min(gx, (tiles + CS_WAVES - 1) / CS_WAVES);
    static DevOnce attr;
    if (attr.first()) { hipFuncSetAttribute((const void*)conv1x1_stream_k<T, EPI, KC, NT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); }
    hipLaunchKernelGGL((conv1x1_stream_k<T, EPI, KC, NT, MT>), dim3(gx, slices), dim3(CS_THREADS), smem, s, a, cg, tiles);
}

template <typename T, int EPI>
static void launch_cs_epi(const GemmArgs& a, const ConvGeom& cg, hipStream_t s) {
    int kc = 0, nt = 0;
    cs_shape(a.K, a.N, &kc, &nt);
    if (kc == 2 && nt == 4) launch_cs<T, EPI, 2, 4>(a, cg, s);
    else if (kc == 2 && nt == 16) launch_cs<T, EPI, 2, 16>(a, cg, s);
    else if (kc == 8 && nt == 4) launch_cs<T, EPI, 8, 4>(a, cg, s);
    else if (kc == 8 && nt == 8) launch_cs<T, EPI, 8, 8>(a, cg, s);
    else if (kc == 4 && nt == 16) launch_cs<T, EPI, 4, 16>(a, cg, s);
}

void launch_conv1x1_stream(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_RELU) launch_cs_epi<T, EPI_RELU>(a, cg, s);
        else if (epi == EPI_RESID_RELU) launch_cs_epi<T, EPI_RESID_RELU>(a, cg, s);
        else launch_cs_epi<T, EPI_NONE>(a, cg, s);
    });
}

}  // namespace rdx

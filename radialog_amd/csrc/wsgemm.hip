// "wsgemm": the encoder's MFMA GEMM / implicit-GEMM convolution for many rows --
//   out[M][N] = epilogue(X[M][K] . W[N][K]^T),  M = batch x pixels (or batch x tokens), K % 64 == 0, N % 16 == 0.
//
// Why another GEMM. gemm_dma_k stages BOTH operands of a 128 x 128 x 64 step through LDS by global_load_lds: 32 KiB of LDS-DMA
// for 128 MFMAs. Measured per shape (tools/enc_kernels.py) every k-step costs ~1 us whatever the grid and however deep the DMA
// ring (a four-stage build was measured in round 2 and removed): the CU's LDS-DMA landing rate (~25 GB/s per loader wave, MI355X_MICROARCH.md "ldsdma-fill") is the limit, and
// the MFMA pipe idles at 20-25 %. Here only the WEIGHT slice goes through LDS (NT x 2 KiB per 64-deep stage, shared by all waves);
// every wave fetches the MFMA B fragments of ITS OWN rows straight into registers (16 rows x 32 k per 16-byte lane load, through
// the vector L1), one stage ahead, and for convolutions computes the im2col address per lane -- no activation bytes in LDS, no
// second wave re-reading them:
//   * workgroup tile = (WAVES x 16 MT rows) x (16 NT columns); wave tile = 16 MT rows x 16 NT columns, NT x MT accumulators;
//   * weight ring of 3 stages in LDS filled by global_load_lds two stages ahead (WB = 2 NT / WAVES one-KiB blocks per wave and
//     stage); activations double-buffered in registers one stage ahead; ONE barrier per stage; the waits on the DMA are counted
//     (`s_waitcnt vmcnt(2 MT + WB)`: in-order return, the younger activation loads and the next DMA stay in flight);
//   * epilogue through a per-wave LDS transpose (the weight ring's space, after a barrier): bias, activation, residual and the
//     store are 16 bytes per lane on full row segments -- same rounding points as everywhere: T(acc + bias), then
//     relu(resid + that).
#include <algorithm>
#include <stdlib.h>

#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

typedef __attribute__((address_space(1))) const void* ws_gptr_t;
typedef __attribute__((address_space(3))) void* ws_lptr_t;

constexpr int WS_ROWB = 144;            // epilogue scratch row pitch: 64 columns x 2 B + 16 B (conflict-free 8-byte column writes)

// SwiGLU on interleaved gate/up tiles (weights.py: rows 0..7 of a 16-row tile = gate, 8..15 = up): T(T(silu(T(g))) * T(u))
template <typename T> __device__ __forceinline__ float ws_swiglu(float gate_acc, float up_acc) {
    const float gt = rnd<T>(gate_acc), up = rnd<T>(up_acc);
    return rnd<T>(silu(gt)) * up;                  // the product is rounded by the store
}

template <typename T, int EPI, int NT, int MT, int WAVES, bool CONV>
__global__ __launch_bounds__(WAVES * 64) void wsgemm_k(GemmArgs a, ConvGeom cg, const void* zero16) {
    typedef typename Vec8<T>::type V8;
    typedef T T4 __attribute__((ext_vector_type(4)));
    constexpr int WB = 2 * NT / WAVES;                          // weight blocks (1 KiB) this wave DMAs per stage
    static_assert(WB >= 1 && WB * WAVES == 2 * NT, "NT / WAVES mismatch");
    constexpr int ROWS = 16 * MT, STAGE = NT * 2 * 64;          // u4 per weight stage
    constexpr bool SILU = (EPI == EPI_SILU_MUL);   // output has N / 2 columns: 8 per 16-column tile
    constexpr int CHN = NT < 4 ? NT : 4, NCH = NT / CHN, LPR = SILU ? CHN : CHN * 2, RPP = 64 / LPR, PASSES = ROWS / RPP;
    extern __shared__ __attribute__((aligned(16))) u4 lds[];    // [3 stages][NT][2 kc][64]   (epilogue: per-wave scratch)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int MB = (a.M + WAVES * ROWS - 1) / (WAVES * ROWS), NB = (a.N + NT * 16 - 1) / (NT * 16);
    const int nwg = MB * NB;
    int tile;
    {   // XCD-aware order: ids that land on one XCD walk consecutive row blocks of one column block (its weight slice stays in that L2)
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int bn = tile / MB, bm = tile - bn * MB;
    const int m0 = (bm * WAVES + w) * ROWS, N0 = bn * NT * 16;
    const int KC = a.K >> 5, NT16 = (a.N + 15) >> 4, nst = a.K >> 6;
    const T* X = reinterpret_cast<const T*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W);

    // weight DMA sources of this wave: block j = (n-tile nt, k-chunk kc) of the stage
    const u4* wsrc[WB];
    int wdst[WB];
#pragma unroll
    for (int j = 0; j < WB; ++j) {
        const int blk = w * WB + j, nt = blk >> 1, kc = blk & 1;
        wsrc[j] = Wp + ((size_t)min((N0 >> 4) + nt, NT16 - 1) * KC + kc) * 64 + lane;
        wdst[j] = blk * 64;
    }
    auto stage_w = [&](int s, int slot) {
#pragma unroll
        for (int j = 0; j < WB; ++j)
            __builtin_amdgcn_global_load_lds((ws_gptr_t)(wsrc[j] + (size_t)s * 2 * 64), (ws_lptr_t)(lds + slot * STAGE + wdst[j]), 16, 0, 0);
    };
    // activation rows of this lane: row m0 + 16 mt + r
    const T* xrow[MT];
    int ih0[MT], iw0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = min(m0 + 16 * mt + r, a.M - 1);
        if (!CONV) {
            xrow[mt] = X + (size_t)row * a.ldx + g * 8;
            ih0[mt] = iw0[mt] = 0;
        } else {
            const int hw = cg.Hout * cg.Wout, b = row / hw, rem = row - b * hw, oh = rem / cg.Wout, ow = rem - oh * cg.Wout;
            xrow[mt] = X + (size_t)b * cg.Hin * cg.Win * cg.Cin + g * 8;
            ih0[mt] = oh * cg.stride - cg.pad;
            iw0[mt] = ow * cg.stride - cg.pad;
        }
    }
    auto load_x = [&](int s, u4 (&xf)[2][MT]) {
        if (!CONV) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) xf[kc][mt] = ldg16(xrow[mt] + (size_t)s * 64 + kc * 32);
        } else {
            // K is ordered (kh, kw, c) and Cin % 64 == 0: a 64-deep stage lies inside ONE filter tap
            const int k = s * 64, kpos = k / cg.Cin, c0 = k - kpos * cg.Cin, kh = kpos / cg.KW, kw = kpos - kh * cg.KW;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ih = ih0[mt] + kh, iw = iw0[mt] + kw;
                const bool inb = ih >= 0 && ih < cg.Hin && iw >= 0 && iw < cg.Win;
                // taps in the zero padding read a 16-byte zero block: the load is ALWAYS issued (the counted vmcnt waits rely on it)
                const T* p = xrow[mt] + ((size_t)ih * cg.Win + iw) * cg.Cin + c0;
                const T* z = reinterpret_cast<const T*>(zero16);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) xf[kc][mt] = ldg16(inb ? p + kc * 32 : z);
            }
        }
    };

    v4f acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (v4f){0.f, 0.f, 0.f, 0.f};

    u4 xa[2][MT], xb[2][MT];
    // issue order per wave (vmcnt is in-order): W(0) X(0) W(1) | step i: X(i+1) W(i+2) ...
    stage_w(0, 0);
    load_x(0, xa);
    stage_w(min(1, nst - 1), 1);
    auto step = [&](int i, u4 (&xcur)[2][MT], u4 (&xnext)[2][MT]) {
        // the DMA of stage i is older than X(i) [2 MT loads] and W(i+1) [WB]: when at most those are outstanding it has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT + WB) : "memory");
        __builtin_amdgcn_s_barrier();                               // stage i visible to all; slot (i+2) % 3 = stage i-1's is free
        load_x(min(i + 1, nst - 1), xnext);
        stage_w(min(i + 2, nst - 1), (i + 2) % 3);
        const u4* base = lds + (i % 3) * STAGE;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const V8 wf = as_vec8<T>(base[(nt * 2 + kc) * 64 + lane]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(wf, as_vec8<T>(xcur[kc][mt]), acc[nt][mt]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of stage i are done before its next barrier
    };
    int i = 0;
    for (; i + 1 < nst; i += 2) {
        step(i, xa, xb);
        step(i + 1, xb, xa);
    }
    if (i < nst) step(i, xa, xb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the clamped tail re-loads must land before the LDS is reused
    __syncthreads();

    // ---- epilogue: MFMA layout (lane = row r, 4 columns at g) -> per-wave scratch [row][64 cols] -> 16-byte row pieces -------------
    unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + (size_t)w * ROWS * WS_ROWB;
    const T* R = reinterpret_cast<const T*>(a.resid);
    T* O = reinterpret_cast<T*>(a.out);
    const int rr = lane / LPR, pc = lane % LPR;
    constexpr bool RES = (EPI == EPI_RESID || EPI == EPI_RESID_RELU);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int nc0 = N0 + c * CHN * 16;                          // first weight row (GEMM column) of the chunk
        if (nc0 >= a.N) break;                                      // (block-uniform) ragged last column block
        const int oc0 = SILU ? (nc0 >> 1) : nc0, OW = SILU ? (a.N >> 1) : a.N;      // output column of the chunk, output width
        u4 rsd[PASSES];
        if (RES) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int m = min(m0 + p * RPP + rr, a.M - 1), n = min(nc0 + pc * 8, a.N - 8);
                rsd[p] = ldg16(R + (size_t)m * a.ldr + n);
            }
        }
#pragma unroll
        for (int q = 0; q < CHN; ++q) {
            const int nt = c * CHN + q, n = min(N0 + nt * 16 + g * 4, a.N - 4);
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) { const float4 bv = *reinterpret_cast<const float4*>(a.bias + n); b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                T4 o;
                if (SILU) {
                    // gate columns sit at g = 0, 1, their up partners at g + 2 = lane ^ 32; the gate lanes write 4 of the tile's 8 outputs
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float up = __shfl_xor(acc[nt][mt][e], 32, 64);
                        o[e] = fromf<T>(ws_swiglu<T>(acc[nt][mt][e], up));
                    }
                    if (g < 2) *reinterpret_cast<T4*>(scr + (16 * mt + r) * WS_ROWB + q * 16 + g * 8) = o;
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[nt][mt][e] + b4[e];
                    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    else if (EPI == EPI_GELU) v = gelu_erf(v);
                    o[e] = fromf<T>(v);
                }
                *reinterpret_cast<T4*>(scr + (16 * mt + r) * WS_ROWB + q * 32 + g * 8) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // one wave, in-order LDS: the chunk is in the scratch
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = p * RPP + rr, m = m0 + row, n = oc0 + pc * 8;
            u4 v = *reinterpret_cast<const u4*>(scr + row * WS_ROWB + pc * 16);
            if (RES) {
                const V8 cv = as_vec8<T>(v), rv = as_vec8<T>(rsd[p]);
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = tof<T>(rv[e]) + tof<T>(cv[e]);
                    if (EPI == EPI_RESID_RELU) s = fmaxf(s, 0.f);
                    o[e] = fromf<T>(s);
                }
                v = as_u4<T>(o);
            }
            if (m < a.M && n + 8 <= OW) stg16(O + (size_t)m * a.ldo + n, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // scratch reads done before the next chunk overwrites it
    }
}

// ---- dispatch -------------------------------------------------------------------------------------------------------------------
// tile shapes: A = 8 waves x 64 rows x 128 columns (512-row workgroup tiles: plain GEMMs with many rows), E = 8 waves x 32 rows x 128
// columns (256-row tiles: the convolutions), D = 8 x 64 rows x 64 columns (N <= 64: layer1), B = 4 waves x 32 rows x 128 columns
// (128-row tiles: 14^2 stage, projector), C = 4 waves x 32 rows x 64 columns (Q-Former at batch: M = 32 x batch rows,
// N = 768 .. 3072). Chosen so that the grid covers the chip.
bool wsgemm_supported(const GemmArgs& a, const ConvGeom& cg, int epi) {
    constexpr int min_rows = 512;
    // (Round 2 also tried it on one prompt's prefill GEMMs -- single 256-row block shapes F / G / H -- and on every encoder shape: prefill of a
    // 160-token prompt 7.5 -> 10.7 ms, long-K / wide-N encoder shapes 10-60 % slower than the 128 x 128 LDS-DMA tiles; both removed in round 3.)
    if (a.M < min_rows || a.K % 64 || a.N % 16 || a.ldo % 8) return false;
    if (!(epi == EPI_NONE || epi == EPI_RELU || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_RESID_RELU)) return false;
    if ((epi == EPI_RESID || epi == EPI_RESID_RELU) && (!a.resid || a.ldr % 8)) return false;
    if (a.N % 8) return false;
    // where it measured faster than gemm_dma_k at batch 32 (tools/enc_kernels.py): the 3x3 convolutions of layer1 / layer2 (Cin <= 128:
    // 112 -> 90, 57 -> 51 us) and the residual epilogues with short K (c3 of layer4 43 -> 33 us, Q-Former output projection 17 -> 12 us: the
    // coalesced epilogue). Long-K, wide-N shapes are bound by the operand bandwidth of a CU either way and the 128 x 128 LDS-DMA tiles do as
    // well or better there.
    if (cg.mode == 1) return cg.Cin % 64 == 0 && cg.Cin <= 128;
    return a.ldx % 8 == 0 && (epi == EPI_RESID || epi == EPI_RESID_RELU) && a.K <= 768;
}

template <typename T, int EPI, int NT, int MT, int WAVES, bool CONV>
static void launch_ws_cfg(const GemmArgs& a, const ConvGeom& cg, const void* zero16, hipStream_t s) {
    const int MB = (a.M + WAVES * 16 * MT - 1) / (WAVES * 16 * MT), NB = (a.N + NT * 16 - 1) / (NT * 16);
    const size_t smem = std::max((size_t)3 * NT * 2 * 1024, (size_t)WAVES * 16 * MT * WS_ROWB);
    static DevOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)wsgemm_k<T, EPI, NT, MT, WAVES, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); }
    hipLaunchKernelGGL((wsgemm_k<T, EPI, NT, MT, WAVES, CONV>), dim3(MB * NB), dim3(WAVES * 64), smem, s, a, cg, zero16);
}

template <typename T, int EPI, bool CONV>
static void launch_ws_epi(const GemmArgs& a, const ConvGeom& cg, const void* zero16, hipStream_t s) {
    const char* e = getenv("RDX_WS_CFG");                        // tests: force a tile shape A .. E
    const int wgA = ((a.M + 511) / 512) * ((a.N + 127) / 128), wgB = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    char cfg = a.N <= 64 ? 'D' : (wgA >= 200 ? 'A' : (wgB >= 160 ? 'B' : 'C'));
    if (e && *e >= 'A' && *e <= 'E') cfg = *e;
    // (the implicit-GEMM address state of a convolution does not fit next to 128 accumulators: its large tile is 256 rows, E)
    if (CONV && cfg == 'A') cfg = 'E';
    if (cfg == 'D') launch_ws_cfg<T, EPI, 4, 4, 8, CONV>(a, cg, zero16, s);          // N <= 64: 512 rows x 64 columns
    else if (cfg == 'E') launch_ws_cfg<T, EPI, 8, 2, 8, CONV>(a, cg, zero16, s);
    else if (cfg == 'A') launch_ws_cfg<T, EPI, 8, 4, 8, false>(a, cg, zero16, s);
    else if (cfg == 'B') launch_ws_cfg<T, EPI, 8, 2, 4, CONV>(a, cg, zero16, s);
    else launch_ws_cfg<T, EPI, 4, 2, 4, CONV>(a, cg, zero16, s);
}

template <typename T, bool CONV>
static void launch_ws_T(const GemmArgs& a, const ConvGeom& cg, int epi, const void* zero16, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: launch_ws_epi<T, EPI_NONE, CONV>(a, cg, zero16, s); break;
        case EPI_RELU: launch_ws_epi<T, EPI_RELU, CONV>(a, cg, zero16, s); break;
        case EPI_GELU: launch_ws_epi<T, EPI_GELU, CONV>(a, cg, zero16, s); break;
        case EPI_RESID: launch_ws_epi<T, EPI_RESID, CONV>(a, cg, zero16, s); break;
        case EPI_RESID_RELU: launch_ws_epi<T, EPI_RESID_RELU, CONV>(a, cg, zero16, s); break;
        default: break;
    }
}

void launch_wsgemm(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, const void* zero16, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, {
        if (cg.mode == 1) launch_ws_T<T, true>(a, cg, epi, zero16, s);
        else launch_ws_T<T, false>(a, cg, epi, zero16, s);
    });
}

}  // namespace rdx

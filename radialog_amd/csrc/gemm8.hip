// fp8 x fp8 MFMA GEMMs of the fp8 weight path (BASELINE configs[4]; gfx950):  out[M,N] = epilogue(Xq[M,K] . Wq[N,K]^T * sx[m][.] * sw[n]).
//
// Weights: OCP e4m3, one absmax / 448 scale per output row, stored in the 64-deep fragment order [n_tile16][k_chunk64][lane (g, r)][16]
// (gemm.hip: pack_weight_fp8_k) -- lane (g, r) holds W[16 nt + r][64 kc + 16 g .. + 16]: one 16-byte piece feeds TWO
// 32-deep MFMA steps (bytes 0..7 and 8..15), or half of a K = 128 / one quarter-K operand of the 32 x 32 x 64 shape. Activations: e4m3 row-major [M][K], quantised per row and per K GROUP
// (`xgroups` ranges of K: group q = 128-deep blocks [NB q / G, NB (q + 1) / G), NB = K / 128; 1 for the projections behind an RMSNorm, 2 for o_proj, 4 for down_proj -- the
// ranges one workgroup of the batch 3-32 K-split decode kernels holds, xstat32.hip, so that one rule describes prefill and decode),
// scale = absmax / 448 in fp32 [M][xgroups] (elem.hip: quant_rows_k, rmsnorm -> fp8). The LoRA-B product and every other epilogue
// stay in the model dtype (finetune.py:167-173 keeps the adapter un-merged, demo.py:232-234); the 2r LoRA-A rows are 16 extra rows of the QKV
// weight and are quantised with it (own row scales).
//   acc (fp32) = sum over the group's k of q_w q_x;  at a group boundary acc *= sx[m][g] / sx[m][g + 1] (rows of this lane; at most three
//   times per GEMM);  out = T(acc * sx[m][last] * sw[n]) -> epilogue. The oracle evaluates sum_g sx[m][g] sum_k q_w q_x sw[n] (fake-quantised
//   operands, fp32): same value up to fp32 rounding order.
// Two kernels, the LDS-DMA pipelines of gemm_dma.hip with twice the MFMAs per staged byte:
//   gemm8_k      128 x 128 block, 4 waves (2 x 2), BK = 128 per step (32 KiB staged: 16 + 16 pieces of 1 KiB), two LDS buffers, counted
//                vmcnt + raw s_barrier; any M, N % 16 == 0, K % 128 == 0;
//   gemm8_256_k  256 (or 320) x 256 block, 8 waves (2 x 4), one 64-deep chunk per stage, four-stage ring (M >= 1024 and >= 256 blocks).
// Both issue the K = 128-rate f8f6f4 instructions (twice the bf16 MFMA rate; v_mfma_f32_16x16x32_fp8_fp8 runs at the bf16 rate) with block scales 2^0:
// gemm8_k v_mfma_f32_16x16x128_f8f6f4 (a step is two chunks; the K groups are whole 128-deep blocks, so a step never straddles a boundary),
// gemm8_256_k v_mfma_f32_32x32x64_f8f6f4 (one chunk per MFMA; see there). Constant-zero scale operands of the builtin select the encoding WITHOUT
// v_mfma_ld_scale_b32 -- hardware scale 1.0 (checked against the fake-quantised fp32 reference, tests/test_gpu_gemm.py) and 5 % faster than loading 0x7F.
// Epilogues: NONE, RESID (out = resid + T(v)), SILU_MUL (gate / up rows interleaved 8 + 8 per tile).
#include <algorithm>
#include <stdlib.h>

#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"      // swiglu()

namespace rdx {

typedef __attribute__((address_space(1))) const void* gptr8_t;
typedef __attribute__((address_space(3))) void* lptr8_t;

// Two consecutive 64-deep pieces of each operand at once on the block-scaled instruction v_mfma_scale_f32_16x16x128_f8f6f4 (formats e4m3 x e4m3,
// every E8M0 block scale = 2^0): one MFMA of K = 128 at TWICE the bf16 rate (MI355X_MICROARCH.md: the non-scaled fp8 16x16x32 runs at the
// bf16 rate). It is a dot product over its 128 k slots, so ANY slot <-> k assignment is valid as long as both operands use the same one:
// a lane's 32 bytes are simply its 16-byte pieces of chunk c and chunk c + 1 (k = 64 c + 16 g .. + 16 and 64 (c + 1) + 16 g .. + 16).
typedef int v8i __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8i pair8(const u4& p0, const u4& p1) {
    return (v8i){(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w, (int)p1.x, (int)p1.y, (int)p1.z, (int)p1.w};
}
__device__ __forceinline__ v4f mfma8x2(const v8i& a, const v8i& b, v4f c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

// The same instruction family in its 32 x 32 x 64 shape: ONE 64-deep chunk of a 32-row tile pair per operand (lane (h, r32) = row r32 of the pair, its two
// 16-byte pieces g = 2 h and 2 h + 1, i.e. k = 32 h .. + 32 of the chunk), 16 fp32 results per lane -- the K = 128 rate on single-chunk stages.
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16f mfma8_32(const v8i& a, const v8i& b, v16f c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

// 4 consecutive columns n .. n + 3 of row m: v already carries both scales
template <typename T, int EPI>
__device__ __forceinline__ void store4_8(const GemmArgs& a, int m, int n, float v[4]) {
    typedef T T4 __attribute__((ext_vector_type(4)));
    T* out = reinterpret_cast<T*>(a.out);
    if (EPI == EPI_RESID) {
        const T4 rv = *reinterpret_cast<const T4*>(reinterpret_cast<const T*>(a.resid) + (size_t)m * a.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tof<T>(rv[e]) + rnd<T>(v[e]);
    }
    T4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fromf<T>(v[e]);
    *reinterpret_cast<T4*>(out + (size_t)m * a.ldo + n) = o;
}

template <typename T, int EPI, int NTW, int MTW>
__device__ __forceinline__ void epilogue8(const GemmArgs& a, v4f (&acc)[NTW][MTW], int M0, int N0, int wm, int wn, int r, int g) {
    const int G = a.xgroups > 0 ? a.xgroups : 1;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = M0 + (wm * MTW + mt) * 16 + r;
        const float sx = a.xscale[(size_t)min(m, a.M - 1) * G + (G - 1)];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = N0 + (wn * NTW + nt) * 16 + g * 4;
            if (n >= a.N) continue;                                   // whole 16-column tile at once (N % 16 == 0)
            const float4 sw = *reinterpret_cast<const float4*>(a.wscale + n);
            float v[4] = {acc[nt][mt][0] * sx * sw.x, acc[nt][mt][1] * sx * sw.y, acc[nt][mt][2] * sx * sw.z, acc[nt][mt][3] * sx * sw.w};
            if (EPI == EPI_SILU_MUL) {
                // rows 0-7 of a tile are gate, 8-15 the matching up rows: the partner (g ^ 2) sits 32 lanes away
                float u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = __shfl_xor(v[e], 32, 64);
                if (g < 2 && m < a.M) {
                    typedef T T4 __attribute__((ext_vector_type(4)));
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fromf<T>(swiglu<T>(v[e], u[e]));
                    const int oc = (N0 >> 1) + (wn * NTW + nt) * 8 + g * 4;
                    *reinterpret_cast<T4*>(reinterpret_cast<T*>(a.out) + (size_t)m * a.ldo + oc) = o;
                }
                continue;
            }
            if (m < a.M) store4_8<T, EPI>(a, m, n, v);
        }
    }
}

// acc *= sx[m][grp] / sx[m][grp + 1] for the rows of this lane (a K-group boundary; wave-uniform branch around it)
template <int NTW, int MTW>
__device__ __forceinline__ void regroup8(const GemmArgs& a, v4f (&acc)[NTW][MTW], int M0, int wm, int r, int grp) {
    const int G = a.xgroups;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = min(M0 + (wm * MTW + mt) * 16 + r, a.M - 1);
        const float ratio = a.xscale[(size_t)m * G + grp] / a.xscale[(size_t)m * G + grp + 1];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][mt][e] *= ratio;
    }
}

constexpr int G8_BM = 128, G8_BN = 128;
constexpr int G8B_NS = 4;                         // LDS ring of the 256-row kernel (a fifth stage, 160 KiB, measured no faster)

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm8_k(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) u4 lds[];          // [stage 2][operand 2][block 16][lane 64]
    const int MB = (a.M + G8_BM - 1) / G8_BM, NB = (a.N + G8_BN - 1) / G8_BN;
    const int nwg = MB * NB;
    int tile;
    {   // XCD-aware order: ids that land on one XCD walk consecutive m-blocks of one n-block (weight panel stays in its L2)
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int bn = tile / MB, bm = tile - bn * MB;
    const int M0 = bm * G8_BM, N0 = bn * G8_BN;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int KC = a.K >> 6, NT16 = (a.N + 15) >> 4;                  // 64-deep chunks
    const int nsteps = KC >> 1;                                        // two chunks per step (K % 128 == 0)
    const unsigned char* X8 = reinterpret_cast<const unsigned char*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W8);

    // this wave stages blocks i = 4 w .. 4 w + 3 of each operand per step; block i = (sub-tile i >> 1, chunk i & 1)
    const u4* wsrc[4];
    const unsigned char* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = w * 4 + j, st = i >> 1, kc = i & 1;
        const int t16 = min((N0 >> 4) + st, NT16 - 1);
        wsrc[j] = Wp + (size_t)t16 * KC * 64 + lane;
        const int row = min(M0 + st * 16 + r, a.M - 1);
        xsrc[j] = X8 + (size_t)row * a.ldx + g * 16;
        (void)kc;
    }
    auto stage = [&](int s, int buf) {
        u4* base = lds + (size_t)buf * 2 * 16 * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = w * 4 + j;
            const int c = min(2 * s + (i & 1), KC - 1);
            __builtin_amdgcn_global_load_lds((gptr8_t)(wsrc[j] + (size_t)c * 64), (lptr8_t)(base + i * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr8_t)(xsrc[j] + (size_t)c * 64), (lptr8_t)(base + (16 + i) * 64), 16, 0, 0);
        }
    };

    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int G = a.xgroups > 0 ? a.xgroups : 1;                      // K group q = steps (128-deep blocks) [nsteps q / G, nsteps (q + 1) / G)
    int grp = 0, next_b = nsteps / G;

    if (nsteps > 0) stage(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) {
            stage(s + 1, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the 8 loads of step s have landed, step s + 1 stays in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const u4* base = lds + (size_t)buf * 2 * 16 * 64;
        if (s == next_b && grp + 1 < G) { regroup8<4, 4>(a, acc, M0, wm, r, grp); ++grp; next_b = (nsteps * (grp + 1)) / G; }
        v8i wf[4], xf[4];                                             // a lane's pieces of both chunks of the step: one K = 128 operand
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {                              // pieces loaded BY VALUE (see gemm8_256_k: the hipcc vmcnt(0) trap)
            const u4 p0 = base[((wn * 4 + nt) * 2 + 0) * 64 + lane], p1 = base[((wn * 4 + nt) * 2 + 1) * 64 + lane];
            wf[nt] = pair8(p0, p1);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const u4 p0 = base[(16 + (wm * 4 + mt) * 2 + 0) * 64 + lane], p1 = base[(16 + (wm * 4 + mt) * 2 + 1) * 64 + lane];
            xf[mt] = pair8(p0, p1);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma8x2(wf[nt], xf[mt], acc[nt][mt]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // everyone is done reading buf before it is re-staged
    }
    epilogue8<T, EPI, 4, 4>(a, acc, M0, N0, wm, wn, r, g);
}

// ---- 256 (or 320) x 256 block: the batched prefill (M = 5120) ---------------------------------------------------------------------------------
// 8 waves as 2 x 4 with 128 (160) x 64 wave tiles = 4 (5) x 2 tiles of 32 x 32 on v_mfma_f32_32x32x64_f8f6f4; gemm_dma256_k's four-stage LDS ring, a stage
// being ONE 64-deep e4m3 chunk of both operands (32 / 36 KiB, [W 16 | X XS sub-tiles][lane (g, r)][16 bytes]). Lane (h, r32) reads row r32 of a sub-tile
// PAIR -- pieces g = 2 h, 2 h + 1 of sub-tile 2 t + (r32 >> 4) -- so one stage is one MFMA per tile and no MFMA needs two resident stages (the K = 128
// shape does: half the DMA lead, measured slower in round 3). D[n = 8 j + 4 h + e][m = r32] (j = reg >> 2, e = reg & 3): four runs of four consecutive
// columns per lane; the SwiGLU partner of a gate row (j even) is reg + 4 of the same lane.
// Round 4 (tools/exp_fp8mx.sh, 32 x 160 tokens): the 16x16x32 fp8 kernel of round 3 took 662 us for gate/up (bf16: 865); this instruction 459; x fragments one
// row-tile pair ahead 437; block scales as constant zeros (no ld_scale) 417 = 2.2 PFLOP/s = 44 % of the fp8 peak; prefill 50.7 -> 37.0 ms. Measured without gain:
// a fifth ring stage (160 KiB), the barrier moved one stage ahead with the next stage's fragments read before it (the code below keeps that form: it needs
// no register copies), all fragments of the next stage in registers before the barrier (128-row tiles), the XCD-compact tile order (halves the fabric
// traffic, 1.75 -> 0.89 GB per gate/up launch, 3 % of time), a K-loop skew between workgroups, four waves of 128 x 128 (761 us). What bounds it: the LDS-DMA ring
// with NOTHING else in the loop takes 366 of the 437 us (3.6 GB of operand requests per launch = 9.85 TB/s out of the L2s, 85 % hits); the matrix pipe is 47 % busy,
// the LDS array 18 %, no bank conflicts (profiles/r04_fp8_prefill_mx.md).
// hipcc detail: an LDS read whose IR load carries no TBAA tag (fragments passed by reference into a helper) gets a conservative s_waitcnt vmcnt(0) in front
// of it while LDS-DMA is in flight -- the ring collapses; loading the pieces BY VALUE (tagged loads) keeps the counted waits below the only ones.
template <int MT32> struct Frag8 { v8i w[2]; v8i x[MT32]; };          // a wave's fragments of one stage: 2 + MT32 operands of 32 bytes per lane

template <typename T, int EPI, int MTW, int NS>
__global__ __launch_bounds__(512) void gemm8_256_k(GemmArgs a) {
    constexpr int XS = 2 * MTW, XPW = (XS + 7) / 8;
    constexpr int LPS = 2 + XPW;
    constexpr int SUB = 16 + XS, BM = XS * 16, MT32 = MTW / 2;
    static_assert(MT32 >= LPS && NS >= 3, "one LDS-DMA piece behind every row-tile pair");
    extern __shared__ __attribute__((aligned(16))) u4 lds[];
    const int MB = (a.M + BM - 1) / BM, NB = (a.N + 255) / 256;
    const int nwg = MB * NB;
    int tile;
    {
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    // The 32 workgroups an XCD runs at a time (consecutive `tile`s) form 4 row blocks x 8 column blocks: 12 operand panels per stage for 32 workgroups in
    // its L2 instead of the 22 of a 20 x 1.6 strip. Bands of 8 column blocks, inside a band groups of 4 row blocks, column-major inside a group.
    const int band = tile / (MB * 8), idx = tile - band * MB * 8, wdt = min(8, NB - band * 8);
    const int gq = idx / (4 * wdt), gh = min(4, MB - gq * 4), rem = idx - gq * 4 * wdt;
    const int bn = band * 8 + rem / gh, bm = gq * 4 + rem % gh;
    const int M0 = bm * BM, N0 = bn * 256;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int h = lane >> 5, r32 = lane & 31, t2 = (lane >> 4) & 1;
    const int wm = w >> 2, wn = w & 3;
    const int KC = a.K >> 6, NT16 = (a.N + 15) >> 4;
    const int nsteps = KC;
    const unsigned char* X8 = reinterpret_cast<const unsigned char*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W8) + lane;

    const u4* wsrc[2];
    const unsigned char* xsrc[XPW];
    int xst[XPW];
#pragma unroll
    for (int j = 0; j < 2; ++j) wsrc[j] = Wp + (size_t)min((N0 >> 4) + w * 2 + j, NT16 - 1) * KC * 64;
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        xst[j] = min(w * XPW + j, XS - 1);
        xsrc[j] = X8 + (size_t)min(M0 + xst[j] * 16 + r, a.M - 1) * a.ldx + g * 16;
    }
    auto stage1 = [&](int s, int slot, int j) {
        u4* base = lds + (size_t)slot * SUB * 64;
        if (j < 2) __builtin_amdgcn_global_load_lds((gptr8_t)(wsrc[j] + (size_t)s * 64), (lptr8_t)(base + (w * 2 + j) * 64), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr8_t)(xsrc[j - 2] + (size_t)s * 64), (lptr8_t)(base + (16 + xst[j - 2]) * 64), 16, 0, 0);
    };
    auto stage = [&](int s, int slot) {
#pragma unroll
        for (int j = 0; j < LPS; ++j) stage1(s, slot, j);
    };

    v16f acc[2][MT32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MT32; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int G = a.xgroups > 0 ? a.xgroups : 1, NBK = KC >> 1;
    int grp = 0, next_b = 2 * (NBK / G);
    // this lane's two pieces of sub-tile pair p of a stage: u4 index (2 p + t2) * 64 + 32 h + (r32 & 15), and 16 further
    const int lpos = t2 * 64 + 32 * h + r;

    // Pipeline: the barrier of iteration s publishes stage s + 1 (and retires stage s - 1, whose slot takes stage s + NS - 1), so the first fragments of
    // stage s + 1 are read behind the last MFMAs of stage s -- no wave ever sits behind a barrier with empty fragment registers.
    auto wait_stage = [&](int younger) {                                 // this wave's loads of all but the `younger` youngest stages have landed
        if (younger * LPS == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger * LPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger * LPS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (younger * LPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger * LPS == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    };
    auto regroup = [&]() {                                                // acc *= sx[m][grp] / sx[m][grp + 1] at a K-group boundary
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt) {
            const int m = min(M0 + (wm * MT32 + mt) * 32 + r32, a.M - 1);
            const float ratio = a.xscale[(size_t)m * G + grp] / a.xscale[(size_t)m * G + grp + 1];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][mt][e] *= ratio;
        }
        ++grp; next_b = 2 * ((NBK * (grp + 1)) / G);
    };
    typedef Frag8<MT32> Frag;
    auto read_w = [&](Frag& f, const u4* base) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { const u4 p0 = base[(wn * 4 + 2 * nt) * 64 + lpos], p1 = base[(wn * 4 + 2 * nt) * 64 + lpos + 16]; f.w[nt] = pair8(p0, p1); }
    };
    auto read_x = [&](Frag& f, const u4* base, int mt) {
        const u4 q0 = base[(16 + wm * MTW + 2 * mt) * 64 + lpos], q1 = base[(16 + wm * MTW + 2 * mt) * 64 + lpos + 16];
        f.x[mt] = pair8(q0, q1);
    };
    // PF = row-tile pairs whose fragments cross the barrier in registers (the rest is read inside the stage, one pair ahead of its MFMAs): all of them
    // when the registers allow (128-row wave tiles: 2 x 48 fragment + 128 accumulator registers), one for the 160-row tiles.
    constexpr int PF = MTW == 8 ? MT32 : 1;
    auto step = [&](int s, Frag& cur, Frag& nxt) {
        wait_stage(NS - 3);                                              // stage s + 1 (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        const int sn = min(s + NS - 1, nsteps - 1), slotn = (s + NS - 1) % NS;
        if (s == next_b && grp + 1 < G) regroup();
        const u4* base = lds + (size_t)(s % NS) * SUB * 64;
        const u4* basen = lds + (size_t)((s + 1) % NS) * SUB * 64;
#pragma unroll
        for (int mt = 0; mt < MT32; ++mt) {
            if (PF < MT32 && mt >= PF - 1 && mt + 1 < MT32) read_x(cur, base, mt + 1);
            if (PF < MT32 && mt == MT32 - 1) { read_w(nxt, basen);
#pragma unroll
                for (int q = 0; q < PF; ++q) read_x(nxt, basen, q); }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][mt] = mfma8_32(cur.w[0], cur.x[mt], acc[0][mt]);
            acc[1][mt] = mfma8_32(cur.w[1], cur.x[mt], acc[1][mt]);
            if (PF == MT32 && mt == 0) {                                 // the whole next stage goes in flight under this stage's remaining MFMAs
                __builtin_amdgcn_sched_barrier(0);
                read_w(nxt, basen);
#pragma unroll
                for (int q = 0; q < MT32; ++q) read_x(nxt, basen, q);
            }
            if (mt < LPS) stage1(sn, slotn, mt);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    Frag fa, fb;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) stage(min(p, nsteps - 1), p);
    wait_stage(NS - 2);
    __builtin_amdgcn_s_barrier();
    read_w(fa, lds);
#pragma unroll
    for (int q = 0; q < PF; ++q) read_x(fa, lds, q);
    for (int s = 0; s < nsteps; s += 2) { step(s, fa, fb); step(s + 1, fb, fa); }   // K % 128 == 0: an even number of stages
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // epilogue: lane (h, r32) holds row m = .. + r32 and, per 32-column tile, the column runs 8 j + 4 h .. + 4 (j = 0 .. 3)
#pragma unroll
    for (int mt = 0; mt < MT32; ++mt) {
        const int m = M0 + (wm * MT32 + mt) * 32 + r32;
        const float sx = a.xscale[(size_t)min(m, a.M - 1) * G + (G - 1)];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int nb = N0 + (wn * 2 + nt) * 32;
            if (EPI == EPI_SILU_MUL) {
#pragma unroll
                for (int tau = 0; tau < 2; ++tau) {               // 16-column tile: 8 gate rows (j = 2 tau), 8 up rows (j = 2 tau + 1)
                    const int n = nb + 16 * tau + 4 * h;
                    if (nb + 16 * tau >= a.N || m >= a.M) continue;
                    const float4 sg = *reinterpret_cast<const float4*>(a.wscale + n), su = *reinterpret_cast<const float4*>(a.wscale + n + 8);
                    const float sgv[4] = {sg.x, sg.y, sg.z, sg.w}, suv[4] = {su.x, su.y, su.z, su.w};
                    typedef T T4 __attribute__((ext_vector_type(4)));
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fromf<T>(swiglu<T>(acc[nt][mt][8 * tau + e] * sx * sgv[e], acc[nt][mt][8 * tau + 4 + e] * sx * suv[e]));
                    const int oc = (nb >> 1) + 8 * tau + 4 * h;
                    *reinterpret_cast<T4*>(reinterpret_cast<T*>(a.out) + (size_t)m * a.ldo + oc) = o;
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nb + 8 * j + 4 * h;
                if (nb + 16 * (j >> 1) >= a.N || m >= a.M) continue;
                const float4 sw = *reinterpret_cast<const float4*>(a.wscale + n);
                float v[4] = {acc[nt][mt][4 * j] * sx * sw.x, acc[nt][mt][4 * j + 1] * sx * sw.y, acc[nt][mt][4 * j + 2] * sx * sw.z, acc[nt][mt][4 * j + 3] * sx * sw.w};
                store4_8<T, EPI>(a, m, n, v);
            }
        }
    }
}

bool gemm8_supported(const GemmArgs& a, int epi) {
    const int G = a.xgroups > 0 ? a.xgroups : 1;
    return a.W8 && a.wscale && a.xscale && a.K % 128 == 0 && a.K / 128 >= G && G <= 4 && a.N % 16 == 0 && a.ldx % 16 == 0 && a.M >= 1 && !a.bias &&
           !a.norm_w && (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_SILU_MUL);
}

template <typename T, int EPI>
static void launch_gemm8_epi(const GemmArgs& a, hipStream_t s) {
    const int MB2 = (a.M + 255) / 256, NB2 = (a.N + 255) / 256;
    if (a.M >= 1024 && a.K >= 512 && a.N >= 1024 && MB2 * NB2 >= 256) {
        // 320-row blocks when they save rounds on the 256 CUs (cost = rounds x rows per block, 15 % handicap for the bigger tile, as gemm_dma256_k)
        const int MB3 = (a.M + 319) / 320;
        const long c256 = (long)((MB2 * NB2 + 255) / 256) * 256, c320 = (long)((MB3 * NB2 + 255) / 256) * 320;
        constexpr size_t sm8 = (size_t)G8B_NS * 32 * 64 * 16, sm10 = (size_t)G8B_NS * 36 * 64 * 16;   // 128 / 144 KiB
        static DevOnce attr;
        if (attr.first()) {
            (void)hipFuncSetAttribute((const void*)gemm8_256_k<T, EPI, 8, G8B_NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
            (void)hipFuncSetAttribute((const void*)gemm8_256_k<T, EPI, 10, G8B_NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm10);
        }
        if (c320 * 115 < c256 * 100) hipLaunchKernelGGL((gemm8_256_k<T, EPI, 10, G8B_NS>), dim3(MB3 * NB2), dim3(512), sm10, s, a);
        else hipLaunchKernelGGL((gemm8_256_k<T, EPI, 8, G8B_NS>), dim3(MB2 * NB2), dim3(512), sm8, s, a);
        return;
    }
    const int MB = (a.M + G8_BM - 1) / G8_BM, NB = (a.N + G8_BN - 1) / G8_BN;
    const size_t smem = (size_t)2 * 2 * 16 * 64 * sizeof(u4);           // 64 KiB
    static DevOnce attr1;
    if (attr1.first()) { hipFuncSetAttribute((const void*)gemm8_k<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); }
    hipLaunchKernelGGL((gemm8_k<T, EPI>), dim3(MB * NB), dim3(256), smem, s, a);
}

void launch_gemm8(int dtype, const GemmArgs& a, int epi, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_RESID) launch_gemm8_epi<T, EPI_RESID>(a, s);
        else if (epi == EPI_SILU_MUL) launch_gemm8_epi<T, EPI_SILU_MUL>(a, s);
        else launch_gemm8_epi<T, EPI_NONE>(a, s);
    });
}

}  // namespace rdx

// fp8 x fp8 MFMA GEMMs of the fp8 weight path (BASELINE configs[4]; gfx950):  out[M,N] = epilogue(Xq[M,K] . Wq[N,K]^T * sx[m][.] * sw[n]).
//
// Weights: OCP e4m3, one absmax / 448 scale per output row, stored in the 64-deep fragment order [n_tile16][k_chunk64][lane (g, r)][16]
// (gemm.hip: pack_weight_fp8_k) -- lane (g, r) holds W[16 nt + r][64 kc + 16 g .. + 16]: one 16-byte piece feeds TWO
// v_mfma_f32_16x16x32_fp8_fp8 (bytes 0..7 and 8..15). Activations: e4m3 row-major [M][K], quantised per row and per K GROUP
// (`xgroups` ranges of K: group q = 64-deep chunks [KC q / G, KC (q + 1) / G); 1 for the projections behind an RMSNorm, 2 for o_proj, 4 for down_proj -- the
// ranges one workgroup of the batch 3-32 K-split decode kernels holds, xstat32.hip, so that one rule describes prefill and decode),
// scale = absmax / 448 in fp32 [M][xgroups] (elem.hip: quant_rows_k, rmsnorm -> fp8). The LoRA-B product and every other epilogue
// stay in the model dtype (finetune.py:167-173 keeps the adapter un-merged, demo.py:232-234).
//   acc (fp32) = sum over the group's k of q_w q_x;  at a group boundary acc *= sx[m][g] / sx[m][g + 1] (rows of this lane; at most three
//   times per GEMM);  out = T(acc * sx[m][last] * sw[n]) -> epilogue. The oracle evaluates sum_g sx[m][g] sum_k q_w q_x sw[n] (fake-quantised
//   operands, fp32): same value up to fp32 rounding order.
// Two kernels, the LDS-DMA pipelines of gemm_dma.hip with twice the MFMAs per staged byte:
//   gemm8_k      128 x 128 block, 4 waves (2 x 2), BK = 128 per step (32 KiB staged: 16 + 16 pieces of 1 KiB), two LDS buffers, counted
//                vmcnt + raw s_barrier; any M, N % 16 == 0, K % 128 == 0 (or K % 64 == 0 with an odd last step handled by clamping);
//   gemm8_256_k  256 x 256 block, 8 waves (2 x 4), one 64-deep chunk per stage, four-stage ring (M >= 1024 and >= 256 blocks).
// Epilogues: NONE, RESID (out = resid + T(v)), SILU_MUL (gate / up rows interleaved 8 + 8 per tile).
#include <algorithm>
#include <stdlib.h>

#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"      // swiglu()

namespace rdx {

typedef __attribute__((address_space(1))) const void* gptr8_t;
typedef __attribute__((address_space(3))) void* lptr8_t;

__device__ __forceinline__ v4f mfma8(const u4& a, const u4& b, v4f c) {
    const long a0 = (long)(((unsigned long long)a.y << 32) | a.x), a1 = (long)(((unsigned long long)a.w << 32) | a.z);
    const long b0 = (long)(((unsigned long long)b.y << 32) | b.x), b1 = (long)(((unsigned long long)b.w << 32) | b.z);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
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
    const int nsteps = (KC + 1) >> 1;                                  // two chunks per step (an odd last chunk is clamped and skipped)
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
    const int G = a.xgroups > 0 ? a.xgroups : 1;                      // K group q = 64-deep chunks [KC q / G, KC (q + 1) / G)
    int grp = 0, next_b = KC / G;

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
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int c = 2 * s + kc;
            if (c >= KC) break;                                       // odd number of chunks: the clamped duplicate is not multiplied
            if (c == next_b && grp + 1 < G) { regroup8<4, 4>(a, acc, M0, wm, r, grp); ++grp; next_b = (KC * (grp + 1)) / G; }
            u4 wf[4], xf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = base[((wn * 4 + nt) * 2 + kc) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = base[(16 + (wm * 4 + mt) * 2 + kc) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma8(wf[nt], xf[mt], acc[nt][mt]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // everyone is done reading buf before it is re-staged
    }
    epilogue8<T, EPI, 4, 4>(a, acc, M0, N0, wm, wn, r, g);
}

// ---- 256 x 256 block, four-stage ring (the batched prefill: M = 5120) ---------------------------------------------------------------
constexpr int G8B_NS = 4, G8B_MTW = 8;

template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm8_256_k(GemmArgs a) {
    constexpr int XS = 16, SUB = 32;                                    // activation sub-tiles / KiB per stage
    extern __shared__ __attribute__((aligned(16))) u4 lds[];          // [stage 4][W 16 | X 16][lane 64]
    const int MB = (a.M + 255) / 256, NB = (a.N + 255) / 256;
    const int nwg = MB * NB;
    int tile;
    {
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int bn = tile / MB, bm = tile - bn * MB;
    const int M0 = bm * 256, N0 = bn * 256;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = w >> 2, wn = w & 3;
    const int KC = a.K >> 6, NT16 = (a.N + 15) >> 4;
    const int nsteps = KC;                                              // one 64-deep chunk per stage
    const unsigned char* X8 = reinterpret_cast<const unsigned char*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W8) + lane;

    // this wave stages weight sub-tiles 2 w, 2 w + 1 and activation sub-tiles 2 w, 2 w + 1
    const u4* wsrc[2];
    const unsigned char* xsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        wsrc[j] = Wp + (size_t)min((N0 >> 4) + w * 2 + j, NT16 - 1) * KC * 64;
        xsrc[j] = X8 + (size_t)min(M0 + (w * 2 + j) * 16 + r, a.M - 1) * a.ldx + g * 16;
    }
    auto stage1 = [&](int s, int slot, int j) {                         // piece j of this wave's 4 pieces of stage s (j compile-time at every call)
        u4* base = lds + (size_t)slot * SUB * 64;
        if (j < 2) __builtin_amdgcn_global_load_lds((gptr8_t)(wsrc[j] + (size_t)s * 64), (lptr8_t)(base + (w * 2 + j) * 64), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr8_t)(xsrc[j - 2] + (size_t)s * 64), (lptr8_t)(base + (16 + w * 2 + j - 2) * 64), 16, 0, 0);
    };
    auto stage = [&](int s, int slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) stage1(s, slot, j);
    };

    v4f acc[4][G8B_MTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < G8B_MTW; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int G = a.xgroups > 0 ? a.xgroups : 1;
    int grp = 0, next_b = KC / G;

#pragma unroll
    for (int p = 0; p < G8B_NS - 1; ++p) stage(min(p, nsteps - 1), p);   // stages 0..2 in flight
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // this wave's 4 loads of stage s have landed (two younger stages may fly)
        __builtin_amdgcn_s_barrier();                                   // ... everyone's have, and everyone finished stage s - 1
        const int sn = min(s + G8B_NS - 1, nsteps - 1), slotn = (s + G8B_NS - 1) % G8B_NS;
        if (s == next_b && grp + 1 < G) { regroup8<4, G8B_MTW>(a, acc, M0, wm, r, grp); ++grp; next_b = (KC * (grp + 1)) / G; }
        const u4* base = lds + (size_t)(s % G8B_NS) * SUB * 64;
        u4 wf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = base[(wn * 4 + nt) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < G8B_MTW; ++mt) {
            const u4 xf = base[(16 + wm * G8B_MTW + mt) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mfma8(wf[nt], xf, acc[nt][mt]);
            // the next stage's LDS-DMA pieces one by one behind the MFMAs of row tiles 1, 3, 5, 7 (see gemm_dma256_k)
            if ((mt & 1) == 1) { stage1(sn, slotn, mt >> 1); __builtin_amdgcn_sched_barrier(0); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // LDS reads of this stage are done before the next barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epilogue8<T, EPI, 4, G8B_MTW>(a, acc, M0, N0, wm, wn, r, g);
}

bool gemm8_supported(const GemmArgs& a, int epi) {
    const int G = a.xgroups > 0 ? a.xgroups : 1;
    return a.W8 && a.wscale && a.xscale && a.K % 64 == 0 && a.K / 64 >= G && G <= 4 && a.N % 16 == 0 && a.ldx % 16 == 0 && a.M >= 1 && !a.bias &&
           !a.norm_w && (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_SILU_MUL);
}

template <typename T, int EPI>
static void launch_gemm8_epi(const GemmArgs& a, hipStream_t s) {
    const int MB2 = (a.M + 255) / 256, NB2 = (a.N + 255) / 256;
    if (a.M >= 1024 && a.K >= 512 && a.N >= 1024 && MB2 * NB2 >= 256) {
        const size_t smem = (size_t)G8B_NS * 32 * 64 * sizeof(u4);      // 128 KiB
        static bool attr = false;
        if (!attr) { hipFuncSetAttribute((const void*)gemm8_256_k<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; }
        hipLaunchKernelGGL((gemm8_256_k<T, EPI>), dim3(MB2 * NB2), dim3(512), smem, s, a);
        return;
    }
    const int MB = (a.M + G8_BM - 1) / G8_BM, NB = (a.N + G8_BN - 1) / G8_BN;
    const size_t smem = (size_t)2 * 2 * 16 * 64 * sizeof(u4);           // 64 KiB
    static bool attr1 = false;
    if (!attr1) { hipFuncSetAttribute((const void*)gemm8_k<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr1 = true; }
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

// Flash-style prefill attention for head_dim 128 (the Llama prompt: causal + key-padding mask), gfx950.
//
// attention_k (attn.hip) gives a workgroup 16 queries, splits their key tiles over its four waves and exchanges the scores through
// LDS ([16][Tk] fp32 + [16][Tk] probabilities): at batch 32 (10 240 workgroups of 10 key tiles) it is bound by VALU issue and by its
// per-workgroup fixed costs -- 107 us per layer for 0.1 us of MFMA work (profiles/r02_bench_b32_kernel_stats.md). Here a WAVE owns 16
// queries for their whole key range, so nothing but V crosses waves and the softmax statistics never leave registers:
//
//   pass 1  S^T = K Q^T per 16-key tile on the matrix cores (A = K fragment, staged once per workgroup and 32-key chunk in LDS in the
//           fragment order; B = the wave's Q fragments; D[key][query]: a lane holds 4 keys of ONE query), scores rounded like the reference, running row maximum and
//           sum of exponentials per lane, combined over the four lane groups of a query once at the end;
//   pass 2  S^T again (the reference rounds the probabilities AFTER normalising by the whole-row sum -- `softmax(..., dtype=float32)
//           .to(query_states.dtype)`, modeling_llama_imgemb.py:229-233 -- so P needs the row statistics first; recomputing 4 MFMAs
//           per tile is cheaper than keeping [64][Tk] scores), P = T(exp(s - m) / l), O^T += V^T P^T.
//           The MFMA k-slot -> key assignment of the P.V product is free (a sum over keys): slot j of lane group g is key
//           16 (j / 4) + 4 g + (j % 4) of the 32-key chunk, which is exactly what the lane already holds from the two score tiles --
//           no cross-lane traffic between the two products. V (stored [key][d]) is staged ONCE per workgroup and chunk, transposed
//           in LDS ([d][key]), so that a lane's A fragment (8 keys of one d) is two 8-byte LDS reads.
//   Rounding points as in attention_k (modeling_llama_imgemb.py:216-234): T(q.k), T(. / sqrt(d)), fp32 softmax (v_exp_f32-based exp and a
//   reciprocal multiply: see fexp), T(p), T(o). Masked keys
//   (padding, causal, beyond Tk) score -inf; a query without any visible key gets zeros (nothing reads such rows).
//
// 64 queries (4 waves) per workgroup: grid (ceil(Tq / 64), heads, batch). Taken when that grid fills the chip (launch_flash_prefill
// decides); one or two prompts keep attention_k, whose many small workgroups suit a latency-bound launch.
#include <stdlib.h>
#include <algorithm>
#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

constexpr int FL_D = 128, FL_DC = FL_D / 32, FL_WAVES = 4, FL_QB = FL_WAVES * 16;
constexpr int FL_VTP = 36;                       // keys per transposed V row in LDS (72 B: the two 4-key halves of a lane land 16 banks apart)

// exp of a non-positive fp32 argument as v_exp_f32(x log2 e). Error bound: the product x log2(e) is rounded once (relative 2^-24, i.e. an
// ABSOLUTE error of |x| log2(e) 2^-24 in the exponent) and v_exp_f32 adds ~1 ulp, so the result is within (1 + |x| 2^-24 ln 2 / 2^-24 ...) =
// about (1 + 0.69 |x|) ulp of expf: 2 ulp near 0, ~8 ulp at x = -10, ~60 ulp at x = -87 (where the probability itself is 1e-38). The fp32
// softmax is rounded to the model dtype right after (2^-8 / 2^-11 relative), so a probability changes -- by one model-dtype ulp -- only when it
// sits within |x| 2^-23 (relative) of a rounding boundary: for the |x| < 16 that carry any weight that is < 2^-19, and attention_k (expf +
// IEEE division) can round such a probability the other way: the two kernels are equal within one model-dtype ulp per probability, NOT bit
// for bit, which is why 1-2 prompts (attention_k) and batched prompts (this kernel) are compared with a tolerance in the tests. expf's range
// handling and the IEEE division of the normalisation were ~40 % of this kernel's vector instructions.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }

template <typename T> __device__ __forceinline__ float scale_score(float s);
// "/ math.sqrt(head_dim)" on a model-dtype value, rounded to the model dtype. bf16: the product with the fp32 reciprocal rounds to the same bf16
// for EVERY finite bf16 input (checked exhaustively against torch's division); fp16 has 52 inputs where it does not, so it divides.
template <> __device__ __forceinline__ float scale_score<bf16>(float s) { return rnd<bf16>(s * 0.08838834764831845f); }
template <> __device__ __forceinline__ float scale_score<f16>(float s) { return rnd<f16>(s / 11.313708498984761f); }

// KPERM (compile time, like the key mask being mandatory): a run-time select between the two K layouts, or a null check of the mask pointer,
// puts every K load of the tile loops into a basic block of its own -- hipcc then branches around each load and waits vmcnt(0) per load
// (cdna_hip_programming.md 5, trap (c)).
// K goes through LDS once per workgroup and chunk, in the MFMA A-fragment order (a 16-key x 32-dim piece is one KiB, written lane-linear, read back
// with one ds_read_b128 per fragment): in the first build every wave fetched its own K fragments from L2 in both passes -- 24 readers of each K
// row per (row, head) against attention_k's 10 -- and ran 106-112 us per layer, no faster than the kernel it was to replace.
template <typename T, bool KPERM>
__global__ __launch_bounds__(FL_WAVES * 64, 4) void flash_prefill_k(AttnArgs a) {
    typedef typename Vec8<T>::type V8;
    typedef T T4 __attribute__((ext_vector_type(4)));
    typedef T T2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) u4 Kl[2][8][64];                 // two chunks of 32 keys: pieces (key tile kt, dim chunk kc) = 4 kt + kc
    __shared__ __attribute__((aligned(16))) T Vt[2][FL_D][FL_VTP];           // two chunks of 32 keys, transposed

    const int q0 = blockIdx.x * FL_QB, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const T* Q = reinterpret_cast<const T*>(a.Q) + b * a.q_bs + h * a.q_hs;
    const T* K = reinterpret_cast<const T*>(a.K) + b * a.k_bs + h * a.k_hs;
    const T* V = reinterpret_cast<const T*>(a.V) + b * a.v_bs + h * a.v_hs;
    const uint8_t* km = a.key_mask + b * a.km_bs;
    const int Tq = a.Tq, Tk = a.Tk, off = Tk - Tq;
    const int qw = q0 + w * 16, q = qw + r;               // this wave's first query, this lane's query

    V8 qf[FL_DC];
#pragma unroll
    for (int kc = 0; kc < FL_DC; ++kc) {
        if (q < Tq) qf[kc] = as_vec8<T>(ldg16(Q + (long)q * a.q_ts + kc * 32 + g * 8));
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kc][j] = fromf<T>(0.f);
        }
    }
    // keys this wave / this workgroup can see (causal: query i attends keys j <= i + off); the workgroup walks whole 32-key chunks
    const int kmax_w = qw < Tq ? min(Tk, qw + 16 + off) : 0;
    const int kmax_g = min(Tk, min(q0 + FL_QB, Tq) + off);
    const int nch = (max(kmax_g, 1) + 31) >> 5;

    // staging roles: wave w fetches K pieces 2 w, 2 w + 1 of a chunk (lane (g, r): 16 bytes of key 16 kt + r, dims 32 kc + 8 g ..) and every thread
    // two 16-byte pieces of V rows (key pair kp, dims d0 ..), written transposed
    const int kp = threadIdx.x >> 4, d0 = (threadIdx.x & 15) * 8;
    auto load_k = [&](int c, u4 (&kr)[2], unsigned (&mw)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * w + j, kt = i >> 2, kc = i & 3;
            const int key = min(c * 32 + kt * 16 + r, Tk - 1);          // keys >= Tk are clamped here and masked in score()
            kr[j] = ldg16(KPERM ? K + kperm(key, kc * 32 + g * 8) : K + (long)key * a.k_ts + kc * 32 + g * 8);
            mw[j] = *reinterpret_cast<const unsigned*>(km + min(c * 32 + j * 16 + g * 4, (int)a.km_bs - 4));     // mask bytes of keys 16 j + 4 g .. + 3
        }
    };
    auto stage_k = [&](int buf, const u4 (&kr)[2]) {
        Kl[buf][2 * w][lane] = kr[0];
        Kl[buf][2 * w + 1][lane] = kr[1];
    };
    auto load_v = [&](int c, u4 (&vr)[2]) {
        const int k0 = min(c * 32 + 2 * kp, Tk - 1), k1 = min(c * 32 + 2 * kp + 1, Tk - 1);   // clamped rows: finite values, probability exactly 0
        vr[0] = ldg16(V + (long)k0 * a.v_ts + d0);
        vr[1] = ldg16(V + (long)k1 * a.v_ts + d0);
    };
    auto stage_v = [&](int buf, const u4 (&vr)[2]) {
        const V8 v0 = as_vec8<T>(vr[0]), v1 = as_vec8<T>(vr[1]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T2 pr; pr[0] = v0[j]; pr[1] = v1[j];
            *reinterpret_cast<T2*>(&Vt[buf][d0 + j][2 * kp]) = pr;
        }
    };
    // the four scores of this lane for key tile kt of chunk c (keys 32 c + 16 kt + 4 g + e, query q): rounded like the reference, -inf where masked
    auto score = [&](int c, int kt, int buf, unsigned mw, float (&sv)[4]) {
        v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < FL_DC; ++kc) acc = mfma16(as_vec8<T>(Kl[buf][kt * 4 + kc][lane]), qf[kc], acc);      // D[i = key_local = 4 g + e][j = q_local = r]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kj = c * 32 + kt * 16 + g * 4 + e;
            const bool ok = (kj < Tk) && (q < Tq) && ((mw >> (8 * e)) & 0xffu) != 0 && kj <= q + off;      // padding, range, causal
            sv[e] = ok ? scale_score<T>(rnd<T>(acc[e])) : -INFINITY;
        }
    };

    // ---- pass 1: row maximum and sum of exponentials ---------------------------------------------------------------------------------
    float m = -INFINITY, l = 0.f;
    {
        u4 kr[2];
        unsigned mnext[2], mcur[2];
        load_k(0, kr, mcur);
        stage_k(0, kr);
        load_k(min(1, nch - 1), kr, mnext);
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            if (c * 32 < kmax_w) {                          // wave-uniform: chunks past this wave's causal range hold no visible key
                float sv[4];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    score(c, kt, c & 1, mcur[kt], sv);
                    // branch-free fold: an all-masked tile leaves (m, l) = (-inf, 0) through exp(-inf) = 0
                    const float mn = fmaxf(m, fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
                    const float ms = mn > -INFINITY ? mn : 0.f;
                    l = l * fexp(m - ms) + ((fexp(sv[0] - ms) + fexp(sv[1] - ms)) + (fexp(sv[2] - ms) + fexp(sv[3] - ms)));
                    m = mn;
                }
            }
            // the next chunk (already in registers) goes to the other buffer, the one after is requested; unconditional (past the end: the last
            // chunk again, into the buffer nobody reads any more), so that no load of the loop sits under a branch
            stage_k((c + 1) & 1, kr);
            mcur[0] = mnext[0]; mcur[1] = mnext[1];
            load_k(min(c + 2, nch - 1), kr, mnext);
            __syncthreads();
        }
    }
    // the four lane groups of a query hold disjoint keys: combine (m, l) over lanes r, r + 16, r + 32, r + 48
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float mo = __shfl_xor(m, o, 64), lo = __shfl_xor(l, o, 64);
        const float mn = fmaxf(m, mo), ms = mn > -INFINITY ? mn : 0.f;
        l = l * fexp(m - ms) + lo * fexp(mo - ms);
        m = mn;
    }
    const bool any_key = (m > -INFINITY) && l > 0.f;
    const float m_use = any_key ? m : 0.f, inv_l = any_key ? 1.0f / l : 0.f;

    // ---- pass 2: P = T(softmax), O^T += V^T P^T over the same chunks; V staged transposed once per workgroup -----------------------------
    v4f acco[FL_D / 16];
#pragma unroll
    for (int i = 0; i < FL_D / 16; ++i) acco[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    {
        u4 kr[2], vr[2];
        unsigned mnext[2], mcur[2];
        load_k(0, kr, mcur);
        load_v(0, vr);
        stage_k(0, kr);
        stage_v(0, vr);
        load_k(min(1, nch - 1), kr, mnext);
        load_v(min(1, nch - 1), vr);
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            const int buf = c & 1;
            if (c * 32 < kmax_w) {
                float s0[4], s1[4];
                score(c, 0, buf, mcur[0], s0);
                score(c, 1, buf, mcur[1], s1);
                V8 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[e] = fromf<T>(fexp(s0[e] - m_use) * inv_l);         // masked: exp(-inf) = 0
                    pf[4 + e] = fromf<T>(fexp(s1[e] - m_use) * inv_l);
                }
#pragma unroll
                for (int dt = 0; dt < FL_D / 16; ++dt) {
                    const T4 lo = *reinterpret_cast<const T4*>(&Vt[buf][dt * 16 + r][4 * g]);
                    const T4 hi = *reinterpret_cast<const T4*>(&Vt[buf][dt * 16 + r][16 + 4 * g]);
                    V8 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                    acco[dt] = mfma16(vf, pf, acco[dt]);             // D[i = d_local = 4 g + e][j = q_local = r]
                }
            }
            stage_k((c + 1) & 1, kr);
            stage_v((c + 1) & 1, vr);
            mcur[0] = mnext[0]; mcur[1] = mnext[1];
            load_k(min(c + 2, nch - 1), kr, mnext);
            load_v(min(c + 2, nch - 1), vr);
            __syncthreads();
        }
    }

    if (q < Tq) {
        T* O = reinterpret_cast<T*>(a.O) + b * a.o_bs + h * a.o_hs;
#pragma unroll
        for (int dt = 0; dt < FL_D / 16; ++dt) {
            T4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fromf<T>(acco[dt][e]);
            if (a.o_packed_mt) {
                // fragment-packed for wstat_k (o_proj): row m = b Tq + q, column k = h D + 16 dt + 4 g .. + 4 (see attention_k)
                const long mrow = (long)b * Tq + q;
                const int k = h * FL_D + dt * 16 + g * 4;
                T* Op = reinterpret_cast<T*>(a.O);
                *reinterpret_cast<T4*>(Op + ((((long)(k >> 5) * a.o_packed_mt + (mrow >> 4)) * 64 + ((k & 31) >> 3) * 16 + (mrow & 15)) << 3) + (k & 7)) = o;
            } else {
                *reinterpret_cast<T4*>(O + (long)q * a.o_ts + dt * 16 + g * 4) = o;
            }
        }
    }
}

bool flash_prefill_supported(int head_dim, const AttnArgs& a) {
    const int min_wgs = a.flash_min;                           // rdx_ctx::flash_min: RDX_FLASH_MIN at rdx_create (default 512), rdx_set_option("flash_min") in tests
    const long wgs = (long)((a.Tq + FL_QB - 1) / FL_QB) * a.H * a.B;
    return min_wgs > 0 && head_dim == FL_D && a.causal && wgs >= min_wgs && (a.v_ts & 7) == 0 && (a.q_ts & 7) == 0 && (a.k_perm || (a.k_ts & 7) == 0) &&
           a.key_mask && (a.km_bs & 3) == 0;
}

void launch_flash_prefill(int dtype, const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.Tq + FL_QB - 1) / FL_QB, a.H, a.B), block(FL_WAVES * 64);
    RDX_DISPATCH_T(dtype, T, {
        if (a.k_perm) hipLaunchKernelGGL((flash_prefill_k<T, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((flash_prefill_k<T, false>), grid, block, 0, s, a);
    });
}

}  // namespace rdx

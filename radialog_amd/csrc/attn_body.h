// Body of decode attention (one new token per batch row) as a device function: runs as its own kernel
// (attn.hip: decode_attention_k, 16 waves) or as the producer role of the fused attention + o_proj launch (fused.hip,
// 8 waves). head_dim 128: one K/V cache row = 256 B = 16 lanes x 16 B.
//
// The cache is small per head (L x 256 B for K and for V), so the work is latency-bound and is organised around three
// workgroup barriers only:
//   * every lane issues ALL its cached K row loads up front (position j -> 16 lanes; PRE x WAVES*4 positions are
//     held in registers, later positions by a second, plain loop); V rows follow as soon as registers allow;
//   * wave 0 alone handles the new token in registers -- LoRA-B add (peft un-merged), rotate-half RoPE (the partner
//     d+-64 lives in lane^8), append to the cache -- and publishes q through LDS;
//   * scores go to LDS once; every wave recomputes the softmax statistics itself (wave shuffles, no block
//     reduction) and applies the rounded probabilities to its V registers.
// Rounding points are the reference's (modeling_llama_imgemb.py:135-142,:198-234): q/k/v in T, T(T(q.k)/sqrt(d)),
// fp32 softmax, P rounded to T before PV, output rounded to T.
#pragma once
#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

template <typename T>
__device__ __forceinline__ float rope_one(float x, float partner_signed, float c, float s) {
    // (x * cos) + (rotate_half(x) * sin), every op rounded to the model dtype
    return rnd<T>(rnd<T>(x * c) + rnd<T>(partner_signed * s));
}

__host__ __device__ inline size_t decode_attention_smem_floats(int waves, int max_len) { return (size_t)waves * 128 + 128 + max_len; }

template <typename T, int WAVES, bool DED = (WAVES == 8)>
__device__ __forceinline__ void decode_attention_body(const DecAttnArgs& a, const int h, const int b, float* dsm) {
    typedef typename Vec8<T>::type V8;
    constexpr int D = 128;
    // 8-wave variant (fused launch, 128-VGPR cap): wave 0 is dedicated to the new token and holds no cache rows, so its
    // register-hungry LoRA/RoPE block never overlaps a live K window; waves 1..7 own the cache rows.
    constexpr int CW = DED ? WAVES - 1 : WAVES;  // waves that own cache rows
    constexpr int SPAN = CW * 4;                 // positions covered by one block-wide load
    constexpr int PRE = DED ? 14 : 8;            // register-resident rows per lane: 512 positions at 16 waves, 392 at 8,
                                                 // 128 at 4 waves (throughput variant for large batches; the rest streams)
    constexpr bool V_EARLY = !DED;               // enough registers to have K and V in flight together
    const LlamaDims& d = a.d;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* lbq = reinterpret_cast<const T*>(a.lbq);
    const T* lbv = reinterpret_cast<const T*>(a.lbv);
    const T* cos_t = reinterpret_cast<const T*>(a.cos_t);
    const T* sin_t = reinterpret_cast<const T*>(a.sin_t);

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* part = dsm;                    // [WAVES][D]
    float* qf = dsm + WAVES * D;          // [D] rotated query
    float* S = dsm + WAVES * D + D;       // [max_len]

    const int jsub = lane >> 4, doct = lane & 15;
    const int H = d.hidden;
    const T* x = qkv + (size_t)b * d.qkv_ld;
    T* kc = reinterpret_cast<T*>(a.kcache) + ((size_t)b * d.heads + h) * d.max_len * D;
    T* vc = reinterpret_cast<T*>(a.vcache) + ((size_t)b * d.heads + h) * d.max_len * D;
    const uint8_t* km = a.key_mask + (size_t)b * d.max_len;

    // ---- cached K rows of this lane in flight first (addresses do not depend on `slot`: rows past it are simply unused)
    const int cw = DED ? w - 1 : w;              // cache-wave index (-1: the dedicated new-token wave)
    const bool owns_rows = !DED || w > 0;        // wave-uniform
    u4 kr[PRE], vr[PRE];
    auto load_k = [&]() {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int j = min(u * SPAN + cw * 4 + jsub, d.max_len - 1);
            kr[u] = ldg16(kc + (size_t)j * D + doct * 8);
        }
    };
    if (!DED) load_k();                          // 16-wave variant: every wave, wave 0 included, owns rows
    const int slot = a.slot_b[b];
    const int nk = slot + 1;

    // ---- new token (wave 0 only, in registers): LoRA add + RoPE for dims doct*8 .. +8 of this head -------------------
    float k8[8], v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k8[e] = 0.f; v8[e] = 0.f; }
    if (w == 0) {
        const int n0 = h * D + doct * 8;
        float q8n[8];
        const int p = a.pos[b];
        const V8 qv = as_vec8<T>(ldg16(x + n0)), kv = as_vec8<T>(ldg16(x + H + n0)), vv = as_vec8<T>(ldg16(x + 2 * H + n0));
        const V8 cv = as_vec8<T>(ldg16(cos_t + (size_t)p * D + doct * 8)), sv_ = as_vec8<T>(ldg16(sin_t + (size_t)p * D + doct * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) { q8n[e] = tof<T>(qv[e]); k8[e] = tof<T>(kv[e]); v8[e] = tof<T>(vv[e]); }
        if (d.lora_r == 8) {
            const V8 aq = as_vec8<T>(ldg16(x + 3 * H)), av = as_vec8<T>(ldg16(x + 3 * H + 8));
#pragma unroll 2
            for (int e = 0; e < 8; ++e) {                        // partial unroll: stays inside the 128-VGPR budget
                const V8 bq = as_vec8<T>(ldg16(lbq + (size_t)(n0 + e) * 8)), bv = as_vec8<T>(ldg16(lbv + (size_t)(n0 + e) * 8));
                float sq = 0.f, sv = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { sq += tof<T>(bq[i]) * tof<T>(aq[i]); sv += tof<T>(bv[i]) * tof<T>(av[i]); }
                q8n[e] = rnd<T>(q8n[e] + rnd<T>(rnd<T>(sq) * d.lora_scale));   // result += lora_B(lora_A(x)) * scaling
                v8[e] = rnd<T>(v8[e] + rnd<T>(rnd<T>(sv) * d.lora_scale));
            }
        }
        const bool lo = doct < 8;                                               // dims < 64: rotate_half gives -x[d+64]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float qo = __shfl_xor(q8n[e], 8, 64), ko = __shfl_xor(k8[e], 8, 64);
            const float c = tof<T>(cv[e]), sn = tof<T>(sv_[e]);
            q8n[e] = rope_one<T>(q8n[e], lo ? -qo : qo, c, sn);
            k8[e] = rope_one<T>(k8[e], lo ? -ko : ko, c, sn);
        }
        if (jsub == 0) {                                                        // publish q, append k / v to the cache
            V8 ko, vo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { qf[doct * 8 + e] = q8n[e]; ko[e] = fromf<T>(k8[e]); vo[e] = fromf<T>(v8[e]); }
            stg16(kc + (size_t)slot * D + doct * 8, as_u4<T>(ko));
            stg16(vc + (size_t)slot * D + doct * 8, as_u4<T>(vo));
        }
    } else if (DED) {
        load_k();                                // exclusive with the new-token block: the K window is never live inside it
    }
    if (V_EARLY && owns_rows) {              // V rows in flight now: their latency hides under the scores and the softmax
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int j = min(u * SPAN + cw * 4 + jsub, d.max_len - 1);
            vr[u] = ldg16(vc + (size_t)j * D + doct * 8);
        }
    }
    __syncthreads();
    float q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q8[e] = qf[doct * 8 + e];

    // ---- scores ---------------------------------------------------------------------------------------------------------
    const float div = sqrtf((float)D);
    auto score_store = [&](int j, float acc) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (doct == 0 && j < nk) {
            const float sc = rnd<T>(rnd<T>(acc) / div);
            S[j] = km[j] ? sc : -INFINITY;
        }
    };
    if (owns_rows) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int j = u * SPAN + cw * 4 + jsub;
            float acc = 0.f;
            if (j < slot) {
                const V8 kv = as_vec8<T>(kr[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += q8[e] * tof<T>(kv[e]);
            }
            score_store(j < slot ? j : nk, acc);                                // one wave-wide call (shuffles inside)
        }
        if (!V_EARLY) {                                                         // K registers are free now
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int j = min(u * SPAN + cw * 4 + jsub, d.max_len - 1);
                vr[u] = ldg16(vc + (size_t)j * D + doct * 8);
            }
        }
    }
    if (w == 0) {                                                               // the new position itself
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += q8[e] * k8[e];
        score_store(jsub == 0 ? slot : nk, acc);
    }
#pragma unroll 4
    for (int j0 = PRE * SPAN; owns_rows && j0 < slot; j0 += SPAN) {             // contexts beyond the register window
        const int j = j0 + cw * 4 + jsub;
        float acc = 0.f;
        if (j < slot) {
            const V8 kv = as_vec8<T>(ldg16(kc + (size_t)j * D + doct * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += q8[e] * tof<T>(kv[e]);
        }
        score_store(j < slot ? j : nk, acc);
    }
    __syncthreads();

    // ---- softmax statistics (fp32), recomputed by every wave -------------------------------------------------------
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, S[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) sum += expf(S[j] - mx);
    sum = wave_sum(sum);

    // ---- O = P V with P rounded to T ---------------------------------------------------------------------------------
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = 0.f;
    if (owns_rows) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int j = u * SPAN + cw * 4 + jsub;
            if (j < slot) {
                const float p = rnd<T>(expf(S[j] - mx) / sum);
                const V8 vv = as_vec8<T>(vr[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] += p * tof<T>(vv[e]);
            }
        }
    }
    if (w == 0 && jsub == 0) {
        const float p = rnd<T>(expf(S[slot] - mx) / sum);
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] += p * v8[e];
    }
    for (int j0 = PRE * SPAN; owns_rows && j0 < slot; j0 += SPAN) {
        const int j = j0 + cw * 4 + jsub;
        if (j < slot) {
            const float p = rnd<T>(expf(S[j] - mx) / sum);
            const V8 vv = as_vec8<T>(ldg16(vc + (size_t)j * D + doct * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] += p * tof<T>(vv[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o8[e] += __shfl_xor(o8[e], 16, 64);
        o8[e] += __shfl_xor(o8[e], 32, 64);
    }
    if (jsub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[w * D + doct * 8 + e] = o8[e];
    }
    __syncthreads();
    if (tid < D) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) v += part[i * D + tid];
        reinterpret_cast<T*>(a.out)[(size_t)b * H + h * D + tid] = fromf<T>(v);
    }
}

}  // namespace rdx

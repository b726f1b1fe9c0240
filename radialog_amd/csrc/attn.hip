// Attention kernels (gfx950).
//
//   attention_k          generic softmax(Q K^T / sqrt(D)) V for short sequences (Llama prefill with causal + padding
//                        mask, Q-Former self/cross attention). QK^T and PV on MFMA 16x16x32, scores of one 16-row
//                        query tile held in LDS, exact two-pass fp32 softmax with the reference's rounding points:
//                        scores rounded to the model dtype, divided by sqrt(D), rounded; fp32 softmax; probabilities
//                        rounded before PV (modeling_llama_imgemb.py:216,:229-234; Qformer.py:195-268 in fp32).
//   rope_kv_prefill_k    LoRA(q,v) add + rotate-half RoPE + KV-cache write for a prefill block
//                        (peft LoRA Linear un-merged, modeling_llama_imgemb.py:135-142,:198-214).
//   decode_attention_k   one new token per row: LoRA + RoPE + in-place KV append + attention over the HBM KV cache.
//                        Bandwidth-bound: K/V rows are streamed once with 16-byte loads (one cache row = 16 lanes),
//                        wavefront-shuffle reductions for the dot products and the softmax.
#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

// ------------------------------------------------------------------------------------------------------------------
// generic attention
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attention_k(AttnArgs a, int TkP) {
    typedef typename Vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* S = reinterpret_cast<float*>(smem);                       // [16][TkP]
    T* P = reinterpret_cast<T*>(smem + (size_t)16 * TkP * 4);        // [16][TkP]

    const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    // wave id made provably wave-uniform: MFMA must never sit under an EXEC-masked (per-lane) branch
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const T* Q = reinterpret_cast<const T*>(a.Q) + b * a.q_bs + h * a.q_hs;
    const T* K = reinterpret_cast<const T*>(a.K) + b * a.k_bs + h * a.k_hs;
    const T* V = reinterpret_cast<const T*>(a.V) + b * a.v_bs + h * a.v_hs;
    T* O = reinterpret_cast<T*>(a.O) + b * a.o_bs + h * a.o_hs;
    const uint8_t* km = a.key_mask ? a.key_mask + b * a.km_bs : nullptr;
    const int Tq = a.Tq, Tk = a.Tk;
    const float inv_div = sqrtf((float)D);
    constexpr int DC = D / 32;

    // ---- phase 1: S = Q K^T ------------------------------------------------------------------------------------
    V8 qf[DC];
    {
        const int q = q0 + r;
#pragma unroll
        for (int kc = 0; kc < DC; ++kc) {
            if (q < Tq) qf[kc] = as_vec8<T>(ldg16(Q + (long)q * a.q_ts + kc * 32 + g * 8));
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[kc][j] = fromf<T>(0.f);
            }
        }
    }
    const int nkt = TkP >> 4;
    for (int kt = w; kt < nkt; kt += 4) {
        const int key = kt * 16 + r;
        v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < DC; ++kc) {
            V8 kf;
            if (key < Tk) kf = as_vec8<T>(ldg16(K + (long)key * a.k_ts + kc * 32 + g * 8));
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) kf[j] = fromf<T>(0.f);
            }
            acc = mfma16(kf, qf[kc], acc);      // D[i = key_local = g*4+e][j = q_local = r]
        }
        const int q = q0 + r;
        float sv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kj = kt * 16 + g * 4 + e;
            bool ok = (kj < Tk) && (q < Tq);
            if (ok && a.causal) ok = kj <= q + (Tk - Tq);
            if (ok && km) ok = km[kj] != 0;
            float s = rnd<T>(acc[e]);           // matmul output in the model dtype
            s = rnd<T>(s / inv_div);            // "/ math.sqrt(head_dim)"
            sv[e] = ok ? s : -INFINITY;
        }
        *reinterpret_cast<float4*>(&S[r * TkP + kt * 16 + g * 4]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    }
    __syncthreads();

    // ---- phase 2: row softmax (fp32), probabilities rounded to T -----------------------------------------------
    {
        const int row = w * 4 + g;              // 16 lanes per row
        float mx = -INFINITY;
        for (int j = r; j < TkP; j += 16) mx = fmaxf(mx, S[row * TkP + j]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        if (mx > -INFINITY) {
            for (int j = r; j < TkP; j += 16) {
                const float e = expf(S[row * TkP + j] - mx);
                S[row * TkP + j] = e;
                sum += e;
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        for (int j = r; j < TkP; j += 16) {
            const float p = (mx > -INFINITY) ? S[row * TkP + j] / sum : 0.f;
            P[row * TkP + j] = fromf<T>(p);
        }
    }
    __syncthreads();

    // ---- phase 3: O = P V -----------------------------------------------------------------------------------------
    const int nkc = TkP >> 5;
    for (int dt = w; dt < D / 16; dt += 4) {
        const int d = dt * 16 + r;
        v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            V8 vf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = kc * 32 + g * 8 + j;
                vf[j] = (key < Tk) ? V[(long)key * a.v_ts + d] : fromf<T>(0.f);
            }
            const V8 pf = *reinterpret_cast<const V8*>(&P[r * TkP + kc * 32 + g * 8]);
            acc = mfma16(vf, pf, acc);          // D[i = d_local = g*4+e][j = q_local = r]
        }
        const int q = q0 + r;
        if (q < Tq) {
            typedef T T4 __attribute__((ext_vector_type(4)));
            T4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fromf<T>(acc[e]);
            *reinterpret_cast<T4*>(O + (long)q * a.o_ts + dt * 16 + g * 4) = o;
        }
    }
}

void launch_attention(int dtype, int head_dim, const AttnArgs& a, hipStream_t s) {
    const int TkP = (a.Tk + 31) & ~31;
    const size_t smem = (size_t)16 * TkP * 6;
    dim3 grid((a.Tq + 15) / 16, a.H, a.B), block(256);
    RDX_DISPATCH_T(dtype, T, {
        if (head_dim == 128) {
            static bool attr128 = false;
            if (!attr128) { hipFuncSetAttribute((const void*)attention_k<T, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr128 = true; }
            hipLaunchKernelGGL((attention_k<T, 128>), grid, block, smem, s, a, TkP);
        } else {
            static bool attr64 = false;
            if (!attr64) { hipFuncSetAttribute((const void*)attention_k<T, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr64 = true; }
            hipLaunchKernelGGL((attention_k<T, 64>), grid, block, smem, s, a, TkP);
        }
    });
}

// ------------------------------------------------------------------------------------------------------------------
// LoRA + RoPE helpers (head_dim 128)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float lora_delta(const T* Bm, int n, const T* avec, int rnk, float scale) {
    // T( T(B . a) * scaling ):  lora_B(lora_A(x)) * scaling with each op rounded to the model dtype
    float acc = 0.f;
    for (int i = 0; i < rnk; ++i) acc += tof<T>(Bm[(size_t)n * rnk + i]) * tof<T>(avec[i]);
    return rnd<T>(rnd<T>(acc) * scale);
}

template <typename T>
__device__ __forceinline__ float rope_one(float x, float partner_signed, float c, float s) {
    // (x * cos) + (rotate_half(x) * sin), every op rounded to the model dtype
    return rnd<T>(rnd<T>(x * c) + rnd<T>(partner_signed * s));
}

template <typename T>
__global__ __launch_bounds__(256) void rope_kv_prefill_k(LlamaDims d, const T* __restrict__ qkv, const T* __restrict__ lbq,
                                                         const T* __restrict__ lbv, const T* __restrict__ cos_t,
                                                         const T* __restrict__ sin_t, const int* __restrict__ pos_ids,
                                                         T* __restrict__ qout, T* __restrict__ kcache,
                                                         T* __restrict__ vcache, int B, int Tn) {
    extern __shared__ float sm[];            // q[hidden], k[hidden]
    float* qs = sm;
    float* ks = sm + d.hidden;
    const int t = blockIdx.x, b = blockIdx.y;
    const size_t row = (size_t)b * Tn + t;
    const T* x = qkv + row * d.qkv_ld;
    const int H = d.hidden, D = d.head_dim;
    const T* aq = x + 3 * H;
    const T* av = x + 3 * H + d.lora_r;
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        float q = tof<T>(x[n]);
        const float k = tof<T>(x[H + n]);
        float v = tof<T>(x[2 * H + n]);
        if (d.lora_r > 0) {
            q = rnd<T>(q + lora_delta<T>(lbq, n, aq, d.lora_r, d.lora_scale));
            v = rnd<T>(v + lora_delta<T>(lbv, n, av, d.lora_r, d.lora_scale));
        }
        qs[n] = q;
        ks[n] = k;
        const int hh = n / D, dd = n - hh * D;
        vcache[(((size_t)b * d.heads + hh) * d.max_len + t) * D + dd] = fromf<T>(v);
    }
    __syncthreads();
    const int pos = pos_ids[row];
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        const int hh = n / D, dd = n - hh * D;
        const float c = tof<T>(cos_t[(size_t)pos * D + dd]), s = tof<T>(sin_t[(size_t)pos * D + dd]);
        const bool lo = dd < D / 2;
        const float qp = lo ? -qs[n + D / 2] : qs[n - D / 2];
        const float kp = lo ? -ks[n + D / 2] : ks[n - D / 2];
        qout[row * H + n] = fromf<T>(rope_one<T>(qs[n], qp, c, s));
        kcache[(((size_t)b * d.heads + hh) * d.max_len + t) * D + dd] = fromf<T>(rope_one<T>(ks[n], kp, c, s));
    }
}

void launch_rope_kv_prefill(int dtype, const LlamaDims& d, const void* qkv, const void* lora_bq, const void* lora_bv,
                            const void* cos_t, const void* sin_t, const int* pos_ids, void* qout, void* kcache,
                            void* vcache, int B, int T_, hipStream_t s) {
    dim3 grid(T_, B), block(256);
    const size_t smem = (size_t)2 * d.hidden * sizeof(float);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rope_kv_prefill_k<T>), grid, block, smem, s, d, (const T*)qkv,
                                                (const T*)lora_bq, (const T*)lora_bv, (const T*)cos_t, (const T*)sin_t,
                                                pos_ids, (T*)qout, (T*)kcache, (T*)vcache, B, T_));
}

// ------------------------------------------------------------------------------------------------------------------
// decode attention (head_dim 128; 16 lanes x 16 B cover one K/V cache row)
// ------------------------------------------------------------------------------------------------------------------
// One workgroup of 16 waves per (head, batch row). The cache is small per head (L x 256 B for K and for V), so the
// kernel is latency-bound and is organised to have only two workgroup barriers:
//   * every lane issues ALL its K and V row loads up front (position j -> 16 lanes; DA_PRE x 64 = 512 positions are
//     prefetched in registers, later positions by a second, plain loop);
//   * the new token's LoRA add + RoPE is done redundantly by every wave in registers (lane (jsub, doct) owns dims
//     doct*8..+8; the rotate-half partner d+-64 lives in lane^8), so no LDS hand-off is needed for q / k_new / v_new;
//   * scores go to LDS once; each wave then recomputes the softmax statistics itself (wave shuffles, no block
//     reduction) and applies the rounded probabilities to its V registers.
// Workgroups with blockIdx.x >= heads (optional) only touch the NEXT GEMV's weights so that they are cache resident
// (the 224 CUs that attention leaves idle at batch 1 pull the o_proj panel towards L2 / Infinity Cache) -- a pure
// performance hint with no data dependence.
constexpr int DA_WAVES = 16;
constexpr int DA_PRE = 8;                     // prefetched (K,V) rows per lane
constexpr int DA_SPAN = DA_WAVES * 4;         // positions covered by one block-wide load

template <typename T>
__global__ __launch_bounds__(DA_WAVES * 64) void decode_attention_k(LlamaDims d, const T* __restrict__ qkv,
                                                                   const T* __restrict__ lbq, const T* __restrict__ lbv,
                                                                   const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                                   const int* __restrict__ pos, const int* __restrict__ slot_b,
                                                                   const uint8_t* __restrict__ key_mask, T* __restrict__ kcache,
                                                                   T* __restrict__ vcache, T* __restrict__ out,
                                                                   const u4* __restrict__ pf_ptr, size_t pf_vec16) {
    typedef typename Vec8<T>::type V8;
    constexpr int D = 128;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= d.heads) {
        // cache-warming role
        if (blockIdx.y != 0 || pf_ptr == nullptr) return;
        const size_t nb = gridDim.x - d.heads, me = blockIdx.x - d.heads;
        unsigned keep = 0;
        for (size_t i = me * blockDim.x + tid; i < pf_vec16; i += nb * blockDim.x) {
            const u4 v = ldg16(pf_ptr + i);
            keep ^= v[0];
        }
        asm volatile("" ::"v"(keep));
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* part = dsm;                    // [DA_WAVES][D]
    float* qf = dsm + DA_WAVES * D;       // [D] rotated query
    float* S = dsm + DA_WAVES * D + D;    // [max_len]

    const int h = blockIdx.x, b = blockIdx.y;
    const int jsub = lane >> 4, doct = lane & 15;
    const int H = d.hidden;
    const int slot = slot_b[b];
    const int nk = slot + 1;
    const T* x = qkv + (size_t)b * d.qkv_ld;
    T* kc = kcache + ((size_t)b * d.heads + h) * d.max_len * D;
    T* vc = vcache + ((size_t)b * d.heads + h) * d.max_len * D;
    const uint8_t* km = key_mask + (size_t)b * d.max_len;

    // ---- all cached K and V rows of this lane in flight first ----------------------------------------------------
    u4 kr[DA_PRE], vr[DA_PRE];
#pragma unroll
    for (int u = 0; u < DA_PRE; ++u) {
        const int j = u * DA_SPAN + w * 4 + jsub;
        const int jc = j < slot ? j : 0;                       // clamped: unconditional loads, unused when j >= slot
        kr[u] = ldg16(kc + (size_t)jc * D + doct * 8);
    }

    // ---- new token (wave 0 only, in registers): LoRA add + RoPE for dims doct*8 .. +8 of this head -------------------
    // lane (jsub, doct) owns 8 dims; the rotate-half partner d+-64 lives in lane^8. q goes to LDS for everybody,
    // k_new / v_new stay in wave 0's registers (it also handles the score / PV term of the new position itself).
    float k8[8], v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k8[e] = 0.f; v8[e] = 0.f; }
    if (w == 0) {
        const int n0 = h * D + doct * 8;
        float q8n[8];
        const V8 qv = as_vec8<T>(ldg16(x + n0)), kv = as_vec8<T>(ldg16(x + H + n0)), vv = as_vec8<T>(ldg16(x + 2 * H + n0));
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
        const int p = pos[b];
        const V8 cv = as_vec8<T>(ldg16(cos_t + (size_t)p * D + doct * 8)), sv_ = as_vec8<T>(ldg16(sin_t + (size_t)p * D + doct * 8));
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
    }

    // V rows go in flight now (their latency hides under the scores and the softmax)
#pragma unroll
    for (int u = 0; u < DA_PRE; ++u) {
        const int j = u * DA_SPAN + w * 4 + jsub;
        const int jc = j < slot ? j : 0;
        vr[u] = ldg16(vc + (size_t)jc * D + doct * 8);
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
#pragma unroll
    for (int u = 0; u < DA_PRE; ++u) {
        const int j = u * DA_SPAN + w * 4 + jsub;
        float acc = 0.f;
        if (j < slot) {
            const V8 kv = as_vec8<T>(kr[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += q8[e] * tof<T>(kv[e]);
        }
        score_store(j < slot ? j : nk, acc);                                    // one wave-wide call (shuffles inside)
    }
    if (w == 0) {                                                               // the new position itself
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += q8[e] * k8[e];
        score_store(jsub == 0 ? slot : nk, acc);
    }
    for (int j0 = DA_PRE * DA_SPAN; j0 < slot; j0 += DA_SPAN) {        // contexts beyond the prefetch window
        const int j = j0 + w * 4 + jsub;
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
#pragma unroll
    for (int u = 0; u < DA_PRE; ++u) {
        const int j = u * DA_SPAN + w * 4 + jsub;
        if (j < slot) {
            const float p = rnd<T>(expf(S[j] - mx) / sum);
            const V8 vv = as_vec8<T>(vr[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] += p * tof<T>(vv[e]);
        }
    }
    if (w == 0 && jsub == 0) {
        const float p = rnd<T>(expf(S[slot] - mx) / sum);
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] += p * v8[e];
    }
    for (int j0 = DA_PRE * DA_SPAN; j0 < slot; j0 += DA_SPAN) {
        const int j = j0 + w * 4 + jsub;
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
        for (int i = 0; i < DA_WAVES; ++i) v += part[i * D + tid];
        out[(size_t)b * H + h * D + tid] = fromf<T>(v);
    }
}

void launch_decode_attention(int dtype, const LlamaDims& d, const void* qkv, const void* lora_bq, const void* lora_bv,
                             const void* cos_t, const void* sin_t, const int* pos, const int* slot_b,
                             const uint8_t* key_mask, void* kcache, void* vcache, void* out, int B, const void* prefetch,
                             size_t prefetch_bytes, hipStream_t s) {
    // at small batch the attention grid leaves most CUs idle: give them the cache-warming role
    const int extra = (prefetch && B * d.heads < 256) ? (256 - B * d.heads > 224 ? 224 : 256 - B * d.heads) : 0;
    dim3 grid(d.heads + extra, B), block(DA_WAVES * 64);
    const size_t smem = (size_t)(DA_WAVES * 128 + 128 + d.max_len) * sizeof(float);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((decode_attention_k<T>), grid, block, smem, s, d, (const T*)qkv,
                                                (const T*)lora_bq, (const T*)lora_bv, (const T*)cos_t, (const T*)sin_t, pos,
                                                slot_b, key_mask, (T*)kcache, (T*)vcache, (T*)out, (const u4*)prefetch,
                                                prefetch_bytes / 16));
}

}  // namespace rdx

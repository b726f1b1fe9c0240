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
#include <stdlib.h>
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "attn_body.h"

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
            if (key < Tk) kf = as_vec8<T>(ldg16(a.k_perm ? K + kperm(key, kc * 32 + g * 8) : K + (long)key * a.k_ts + kc * 32 + g * 8));
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
            if (a.o_packed_mt) {
                // fragment-packed for wstat_k (o_proj): row m = b Tq + q, column k = h D + 16 dt + 4 g .. + 4 -> half of the 16-byte piece of
                // lane (g' = (k % 32) / 8, r' = m % 16) in fragment (k / 32, m / 16)
                const long m = (long)b * Tq + q;
                const int k = h * D + dt * 16 + g * 4;
                T* Op = reinterpret_cast<T*>(a.O);
                *reinterpret_cast<T4*>(Op + ((((long)(k >> 5) * a.o_packed_mt + (m >> 4)) * 64 + ((k & 31) >> 3) * 16 + (m & 15)) << 3) + (k & 7)) = o;
            } else
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
        } else if (head_dim == 32) {
            static bool attr32 = false;
            if (!attr32) { hipFuncSetAttribute((const void*)attention_k<T, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr32 = true; }
            hipLaunchKernelGGL((attention_k<T, 32>), grid, block, smem, s, a, TkP);
        } else {
            static bool attr64 = false;
            if (!attr64) { hipFuncSetAttribute((const void*)attention_k<T, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr64 = true; }
            hipLaunchKernelGGL((attention_k<T, 64>), grid, block, smem, s, a, TkP);
        }
    });
}

// ------------------------------------------------------------------------------------------------------------------
// LoRA + RoPE + KV-cache write of the prompt (head_dim 128)
// ------------------------------------------------------------------------------------------------------------------
// one thread = 8 contiguous dims of one (token, head): 16-byte loads/stores throughout; the rotate-half partner of q
// (which needs the LoRA-updated value) is exchanged through LDS, the partner of k is read straight from the GEMM output.
template <typename T>
__global__ __launch_bounds__(1024) void rope_kv_prefill_k(LlamaDims d, const T* __restrict__ qkv, const T* __restrict__ lbq,
                                                          const T* __restrict__ lbv, const T* __restrict__ cos_t,
                                                          const T* __restrict__ sin_t, const int* __restrict__ pos_ids,
                                                          T* __restrict__ qout, T* __restrict__ kcache,
                                                          T* __restrict__ vcache, int B, int Tn, int slot0) {
    typedef typename Vec8<T>::type V8;
    extern __shared__ float qs[];            // [hidden] q after the LoRA add
    constexpr int D = 128;
    const int t = blockIdx.x, b = blockIdx.y;
    const size_t row = (size_t)b * Tn + t;
    const T* x = qkv + row * d.qkv_ld;
    const int H = d.hidden;
    const int n0 = threadIdx.x * 8;
    const bool act = n0 < H;
    const int hh = n0 / D, dd = n0 - hh * D;
    const bool lo = dd < D / 2;
    float q8[8], k8[8], kp8[8];
    if (act) {
        const V8 qv = as_vec8<T>(ldg16(x + n0)), kv = as_vec8<T>(ldg16(x + H + n0)), vv = as_vec8<T>(ldg16(x + 2 * H + n0));
        const V8 kpv = as_vec8<T>(ldg16(x + H + (lo ? n0 + D / 2 : n0 - D / 2)));
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { q8[e] = tof<T>(qv[e]); k8[e] = tof<T>(kv[e]); v8[e] = tof<T>(vv[e]); kp8[e] = tof<T>(kpv[e]); }
        if (d.lora_r == 8) {
            const V8 aq = as_vec8<T>(ldg16(x + 3 * H)), av = as_vec8<T>(ldg16(x + 3 * H + 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const V8 bq = as_vec8<T>(ldg16(lbq + (size_t)(n0 + e) * 8)), bv = as_vec8<T>(ldg16(lbv + (size_t)(n0 + e) * 8));
                float sq = 0.f, sv = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { sq += tof<T>(bq[i]) * tof<T>(aq[i]); sv += tof<T>(bv[i]) * tof<T>(av[i]); }
                q8[e] = rnd<T>(q8[e] + rnd<T>(rnd<T>(sq) * d.lora_scale));     // result += lora_B(lora_A(x)) * scaling
                v8[e] = rnd<T>(v8[e] + rnd<T>(rnd<T>(sv) * d.lora_scale));
            }
        }
        V8 vo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { qs[n0 + e] = q8[e]; vo[e] = fromf<T>(v8[e]); }
        stg16(vcache + (((size_t)b * d.heads + hh) * d.max_len + slot0 + t) * D + dd, as_u4<T>(vo));
    }
    __syncthreads();
    if (act) {
        const int pos = pos_ids[row];
        const V8 cv = as_vec8<T>(ldg16(cos_t + (size_t)pos * D + dd)), sv = as_vec8<T>(ldg16(sin_t + (size_t)pos * D + dd));
        const int pn = lo ? n0 + D / 2 : n0 - D / 2;
        V8 qo, ko;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = tof<T>(cv[e]), sn = tof<T>(sv[e]);
            const float qp = lo ? -qs[pn + e] : qs[pn + e];
            const float kp = lo ? -kp8[e] : kp8[e];
            qo[e] = fromf<T>(rope_one<T>(q8[e], qp, c, sn));
            ko[e] = fromf<T>(rope_one<T>(k8[e], kp, c, sn));
        }
        stg16(qout + row * H + n0, as_u4<T>(qo));
        stg16(kcache + ((size_t)b * d.heads + hh) * d.max_len * D + kperm(slot0 + t, dd, d.k_perm), as_u4<T>(ko));     // fragment order per 16 positions
    }
}

void launch_rope_kv_prefill(int dtype, const LlamaDims& d, const void* qkv, const void* lora_bq, const void* lora_bv,
                            const void* cos_t, const void* sin_t, const int* pos_ids, void* qout, void* kcache,
                            void* vcache, int B, int T_, int slot0, hipStream_t s) {
    const int threads = ((d.hidden / 8 + 63) / 64) * 64;          // hidden <= 8192
    dim3 grid(T_, B), block(threads);
    const size_t smem = (size_t)d.hidden * sizeof(float);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rope_kv_prefill_k<T>), grid, block, smem, s, d, (const T*)qkv,
                                                (const T*)lora_bq, (const T*)lora_bv, (const T*)cos_t, (const T*)sin_t,
                                                pos_ids, (T*)qout, (T*)kcache, (T*)vcache, B, T_, slot0));
}

// test introspection: the K cache of one layer back in [B][heads][max_len][128] row-major order
__global__ void k_unperm_k(const unsigned short* __restrict__ kc, unsigned short* __restrict__ out, int max_len, size_t total8, int perm) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece
    if (i >= total8) return;
    const size_t slab = i / ((size_t)max_len * 16), rem = i - slab * (size_t)max_len * 16;
    const int pos = (int)(rem >> 4), dim = (int)(rem & 15) * 8;
    stg16(out + slab * max_len * 128 + (size_t)pos * 128 + dim, ldg16(kc + slab * max_len * 128 + kperm(pos, dim, perm)));
}
void launch_k_unperm(const void* kc, void* out, size_t slabs, int max_len, int perm, hipStream_t s) {
    const size_t total8 = slabs * max_len * 16;
    hipLaunchKernelGGL(k_unperm_k, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, (const unsigned short*)kc, (unsigned short*)out, max_len, total8, perm);
}

// ------------------------------------------------------------------------------------------------------------------
// decode attention (body in attn_body.h)
// ------------------------------------------------------------------------------------------------------------------
constexpr int DA_WAVES = 16;       // latency variant: few (head, row) pairs, each gets a whole CU
constexpr int DA_WAVES_TP = 4;     // throughput variant: > 256 pairs, 4 workgroups per CU share the KV stream (8 waves with a 256-position
                                   // register window measured 5 % slower at batch 32)

template <typename T, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void decode_attention_k(DecAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    // latency variant: wave 0 is dedicated to the new token (its operand loads are first in its queue), 15 waves own the cache
    if (WAVES == 16) decode_attention_body<T, WAVES, true, NoWait, true>(a, blockIdx.x, blockIdx.y, dsm);
    else decode_attention_body<T, WAVES, false>(a, blockIdx.x, blockIdx.y, dsm);
}

void launch_decode_attention(int dtype, const DecAttnArgs& a, int B, hipStream_t s) {
    dim3 grid(a.d.heads, B);
    const char* tp_env = getenv("RDX_ATT_TP");                       // tests: force the throughput variant
    const int force_tp = tp_env ? atoi(tp_env) : 0;
    if (a.d.heads * B <= 256 && !force_tp) {
        const size_t smem = decode_attention_smem_floats(DA_WAVES, a.d.max_len) * sizeof(float);
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((decode_attention_k<T, DA_WAVES>), grid, dim3(DA_WAVES * 64), smem, s, a));
    } else {
        const size_t smem = decode_attention_smem_floats(DA_WAVES_TP, a.d.max_len) * sizeof(float);
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((decode_attention_k<T, DA_WAVES_TP>), grid, dim3(DA_WAVES_TP * 64), smem, s, a));
    }
}

}  // namespace rdx

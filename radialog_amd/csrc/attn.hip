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
#include <algorithm>
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
    T* P = reinterpret_cast<T*>(smem);                               // [16][TkP]
    float* S = reinterpret_cast<float*>(smem + (size_t)16 * TkP * 2);   // [16][TkP], dead after phase 2 ...
    T* Vt_all = reinterpret_cast<T*>(smem + (size_t)16 * TkP * 2);   // ... when the waves' transposed V patches [4][2 d tiles][16 d][VTP keys] take its place

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
    // causal: this block of 16 queries sees keys <= q0 + 15 + (Tk - Tq) -- later key tiles are neither scored nor multiplied (at Tq = Tk half of
    // all tiles). TkE = the keys this block works on, rounded up to a k-chunk of 32; keys in [kmax, TkE) are masked like any other
    const int kmax = a.causal ? min(Tk, q0 + 16 + (Tk - Tq)) : Tk;
    const int TkE = min(TkP, (max(kmax, 1) + 31) & ~31);
    const int nkt = TkE >> 4;
    // K fragments one key tile ahead: the loop is a chain of dependent L2 round trips otherwise (a block's time was three of them here plus five in
    // phase 3 -- 15 us for 0.1 us of MFMA work). Keys >= Tk are clamped (their scores are masked below).
    // ... and with them the tile's key-mask bytes, one 32-bit word per lane (keys 16 kt + 4 g .. + 3) instead of four byte loads behind the MFMAs
    auto load_kt = [&](int kt, V8 (&kf)[DC], unsigned& mw) {
        const int key = min(kt * 16 + r, Tk - 1);
#pragma unroll
        for (int kc = 0; kc < DC; ++kc)
            kf[kc] = as_vec8<T>(ldg16(a.k_perm ? K + kperm(key, kc * 32 + g * 8) : K + (long)key * a.k_ts + kc * 32 + g * 8));
        mw = km ? *reinterpret_cast<const unsigned*>(km + min(kt * 16 + g * 4, (int)a.km_bs - 4)) : 0x01010101u;   // clamped only where kj >= Tk
    };
    auto score = [&](int kt, const V8 (&kf)[DC], unsigned mw) {
        v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < DC; ++kc) acc = mfma16(kf[kc], qf[kc], acc);      // D[i = key_local = g*4+e][j = q_local = r]
        const int q = q0 + r;
        float sv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kj = kt * 16 + g * 4 + e;
            bool ok = (kj < Tk) && (q < Tq);
            if (ok && a.causal) ok = kj <= q + (Tk - Tq);
            ok = ok && ((mw >> (8 * e)) & 0xffu) != 0;
            float s = rnd<T>(acc[e]);           // matmul output in the model dtype
            s = rnd<T>(s / inv_div);            // "/ math.sqrt(head_dim)"
            sv[e] = ok ? s : -INFINITY;
        }
        *reinterpret_cast<float4*>(&S[r * TkP + kt * 16 + g * 4]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    };
    // two register sets used in turn (no copy between them: a copy of the set just requested would wait for it); the request past the last tile
    // re-reads the last one
    V8 ka[DC], kb[DC];
    unsigned ma = 0, mb = 0;
    {
        int kt = w;
        if (kt < nkt) load_kt(kt, ka, ma);
        while (kt < nkt) {
            load_kt(min(kt + 4, nkt - 1), kb, mb);
            score(kt, ka, ma);
            kt += 4;
            if (kt >= nkt) break;
            load_kt(min(kt + 4, nkt - 1), ka, ma);
            score(kt, kb, mb);
            kt += 4;
        }
    }
    __syncthreads();

    // ---- phase 2: row softmax (fp32), probabilities rounded to T -----------------------------------------------
    {
        const int row = w * 4 + g;              // 16 lanes per row
        float mx = -INFINITY;
        for (int j = r; j < TkE; j += 16) mx = fmaxf(mx, S[row * TkP + j]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        if (mx > -INFINITY) {
            for (int j = r; j < TkE; j += 16) {
                const float e = expf(S[row * TkP + j] - mx);
                S[row * TkP + j] = e;
                sum += e;
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        for (int j = r; j < TkE; j += 16) {
            const float p = (mx > -INFINITY) ? S[row * TkP + j] / sum : 0.f;
            P[row * TkP + j] = fromf<T>(p);
        }
    }
    __syncthreads();

    // ---- phase 3: O = P V -----------------------------------------------------------------------------------------
    // V is [key][d] in memory and the MFMA A fragment wants 8 consecutive KEYS of one d per lane. Each wave therefore stages the [32 keys][16 d]
    // block of a k-chunk with ONE 16-byte load per lane (lane = (key, half of the d tile): 32 B of a V row per key, coalesced across the waves'
    // d tiles), writes it transposed into its own LDS patch and reads the fragment back with two ds_read_b64. (Eight 2-byte global loads per lane and
    // fragment kept the texture-address unit busy for 138 us per batch-32 prefill layer.) Wave-private patch: LDS serves a wave's operations in order.
    constexpr int VTP = 36;                                          // keys per transposed row (72 B: the two halves of a tile land 16 banks apart)
    constexpr int NDW = (D / 16 + 3) / 4;                            // d tiles per wave: 2 (D = 128), 1 (64, 32)
    const int nkc = TkE >> 5;
    T* Vt = Vt_all + (size_t)w * 2 * 16 * VTP;
    const bool vvec = (a.v_ts & 7) == 0;                             // 16-byte V loads need 8-element row strides (every caller has them)
    v4f accd[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) accd[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int vkey = lane >> 1, vhalf = lane & 1;
    // V one k-chunk ahead (same reason as the K fragments of phase 1)
    auto load_v = [&](int kc, u4 (&vr)[NDW]) {
        const int key = min(kc * 32 + vkey, Tk - 1);     // keys >= Tk: a clamped (valid, finite) row -- its probability is exactly 0 (masked in phase 1)
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const int dt = min(w + 4 * i, D / 16 - 1);
            vr[i] = ldg16(V + (long)key * a.v_ts + dt * 16 + vhalf * 8);
        }
    };
    auto pv = [&](int kc, const u4 (&vr)[NDW]) {
        const V8 pf = *reinterpret_cast<const V8*>(&P[r * TkP + kc * 32 + g * 8]);
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const int dt = w + 4 * i;
            if (dt < D / 16) {                                       // wave-uniform
                V8 vf;
                if (vvec) {
                    const V8 vv = as_vec8<T>(vr[i]);
                    T* col = Vt + ((size_t)i * 16 + vhalf * 8) * VTP + vkey;
#pragma unroll
                    for (int j = 0; j < 8; ++j) col[(size_t)j * VTP] = vv[j];
                    typedef T T4v __attribute__((ext_vector_type(4)));
                    const T4v* rp = reinterpret_cast<const T4v*>(Vt + ((size_t)i * 16 + r) * VTP + g * 8);
                    const T4v lo = rp[0], hi = rp[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                } else {
                    const int d = dt * 16 + r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = kc * 32 + g * 8 + j;
                        vf[j] = (key < Tk) ? V[(long)key * a.v_ts + d] : fromf<T>(0.f);
                    }
                }
                accd[i] = mfma16(vf, pf, accd[i]);          // D[i = d_local = g*4+e][j = q_local = r]
            }
        }
    };
    u4 va[NDW], vb[NDW];
    if (vvec) {
        int kc = 0;
        if (kc < nkc) load_v(kc, va);
        while (kc < nkc) {
            load_v(min(kc + 1, nkc - 1), vb);
            pv(kc, va);
            if (++kc >= nkc) break;
            load_v(min(kc + 1, nkc - 1), va);
            pv(kc, vb);
            ++kc;
        }
    } else {
        for (int kc = 0; kc < nkc; ++kc) pv(kc, va);
    }
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        const int dt = w + 4 * i;
        if (dt >= D / 16) continue;
        const v4f acc = accd[i];
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
    // many prompts of head_dim 128 (the batched Llama prefill): 64-query flash-style blocks (flash.hip)
    if (flash_prefill_supported(head_dim, a)) { launch_flash_prefill(dtype, a, s); return; }
    const int TkP = (a.Tk + 31) & ~31;
    const size_t smem = (size_t)16 * TkP * 2 + std::max((size_t)16 * TkP * 4, (size_t)4 * 2 * 16 * 36 * 2);     // P, then S / the waves' transposed V patches
    dim3 grid((a.Tq + 15) / 16, a.H, a.B), block(256);
    RDX_DISPATCH_T(dtype, T, {
        if (head_dim == 128) {
            static DevOnce attr128;
            if (attr128.first()) { hipFuncSetAttribute((const void*)attention_k<T, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
            hipLaunchKernelGGL((attention_k<T, 128>), grid, block, smem, s, a, TkP);
        } else if (head_dim == 32) {
            static DevOnce attr32;
            if (attr32.first()) { hipFuncSetAttribute((const void*)attention_k<T, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
            hipLaunchKernelGGL((attention_k<T, 32>), grid, block, smem, s, a, TkP);
        } else {
            static DevOnce attr64;
            if (attr64.first()) { hipFuncSetAttribute((const void*)attention_k<T, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
            hipLaunchKernelGGL((attention_k<T, 64>), grid, block, smem, s, a, TkP);
        }
    });
}

// ------------------------------------------------------------------------------------------------------------------
// LoRA + RoPE + KV-cache write of the prompt (head_dim 128)
// ------------------------------------------------------------------------------------------------------------------
// one thread = 8 contiguous dims of one (token, head): 16-byte loads/stores throughout; the rotate-half partner of q
// (which needs the LoRA-updated value) sits 8 lanes away in the same 16-lane row (a head = 16 threads) and comes by DPP -- round 3: it used
// to go through LDS behind two workgroup barriers per token, which serialised the eight waves of a workgroup on every token's load latency
// (62 us per layer at 32 x 160 tokens for 250 MB); the partner of k is read straight from the GEMM output.
// A workgroup handles `tpb` consecutive tokens of a prompt: a thread's LoRA-B rows (8 rows x 16 B for q and for v: 256 B per thread, 128 KiB per
// workgroup) are read ONCE and stay in registers -- with one token per workgroup the batched prefill re-read them for each of its 5120 tokens
// (655 MB through L2 per layer, 108 us against the 50 us its 250 MB of QKV / q / K / V traffic needs).
template <typename T>
__global__ __launch_bounds__(1024) void rope_kv_prefill_k(LlamaDims d, const T* __restrict__ qkv, const T* __restrict__ lbq,
                                                          const T* __restrict__ lbv, const T* __restrict__ cos_t,
                                                          const T* __restrict__ sin_t, const int* __restrict__ pos_ids,
                                                          T* __restrict__ qout, T* __restrict__ kcache,
                                                          T* __restrict__ vcache, int B, int Tn, int slot0, int tpb) {
    typedef typename Vec8<T>::type V8;
    constexpr int D = 128;
    const int b = blockIdx.y;
    const int H = d.hidden;
    const int n0 = threadIdx.x * 8;
    const bool act = n0 < H;
    const int hh = n0 / D, dd = n0 - hh * D;
    const bool lo = dd < D / 2;
    const bool lora = d.lora_r == 8;
    u4 bqr[8], bvr[8];                       // the rows stay PACKED (64 VGPRs): see the empty asm in the loop
#pragma unroll
    for (int e = 0; e < 8; ++e) { bqr[e] = (u4){0u, 0u, 0u, 0u}; bvr[e] = (u4){0u, 0u, 0u, 0u}; }
    if (act && lora) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { bqr[e] = ldg16(lbq + (size_t)(n0 + e) * 8); bvr[e] = ldg16(lbv + (size_t)(n0 + e) * 8); }
    }
    for (int t = blockIdx.x * tpb; t < min((int)(blockIdx.x + 1) * tpb, Tn); ++t) {
        // opaque to the optimiser: without it the 128 conversions to fp32 are hoisted out of the token loop and the kernel needs 204 VGPRs
        // (one workgroup per CU); packed, it fits 128 and two workgroups share a CU
        V8 bq[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            asm volatile("" : "+v"(bqr[e].x), "+v"(bqr[e].y), "+v"(bqr[e].z), "+v"(bqr[e].w));
            asm volatile("" : "+v"(bvr[e].x), "+v"(bvr[e].y), "+v"(bvr[e].z), "+v"(bvr[e].w));
            bq[e] = as_vec8<T>(bqr[e]); bv[e] = as_vec8<T>(bvr[e]);
        }
        const size_t row = (size_t)b * Tn + t;
        const T* x = qkv + row * d.qkv_ld;
        float q8[8], k8[8], kp8[8];
        if (act) {
            const V8 qv = as_vec8<T>(ldg16(x + n0)), kv = as_vec8<T>(ldg16(x + H + n0)), vv = as_vec8<T>(ldg16(x + 2 * H + n0));
            const V8 kpv = as_vec8<T>(ldg16(x + H + (lo ? n0 + D / 2 : n0 - D / 2)));
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { q8[e] = tof<T>(qv[e]); k8[e] = tof<T>(kv[e]); v8[e] = tof<T>(vv[e]); kp8[e] = tof<T>(kpv[e]); }
            if (lora) {
                const V8 aq = as_vec8<T>(ldg16(x + 3 * H)), av = as_vec8<T>(ldg16(x + 3 * H + 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float sq = 0.f, sv = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { sq += tof<T>(bq[e][i]) * tof<T>(aq[i]); sv += tof<T>(bv[e][i]) * tof<T>(av[i]); }
                    q8[e] = rnd<T>(q8[e] + rnd<T>(rnd<T>(sq) * d.lora_scale));     // result += lora_B(lora_A(x)) * scaling
                    v8[e] = rnd<T>(v8[e] + rnd<T>(rnd<T>(sv) * d.lora_scale));
                }
            }
            V8 vo;
#pragma unroll
            for (int e = 0; e < 8; ++e) vo[e] = fromf<T>(v8[e]);
            stg16(vcache + (((size_t)b * d.heads + hh) * d.max_len + slot0 + t) * D + dd, as_u4<T>(vo));
        }
        // the DPP exchange runs for every lane of the wave (whole heads are active or inactive together: hidden % 128 == 0)
        float qp8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qp8[e] = dpp_mov<DPP_ROR8>(act ? q8[e] : 0.f);          // lane ^ 8 inside the 16-lane row: dims +- 64 of the same head
        if (act) {
            const int pos = pos_ids[row];
            const V8 cv = as_vec8<T>(ldg16(cos_t + (size_t)pos * D + dd)), sv = as_vec8<T>(ldg16(sin_t + (size_t)pos * D + dd));
            V8 qo, ko;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = tof<T>(cv[e]), sn = tof<T>(sv[e]);
                const float qp = lo ? -qp8[e] : qp8[e];
                const float kp = lo ? -kp8[e] : kp8[e];
                qo[e] = fromf<T>(rope_one<T>(q8[e], qp, c, sn));
                ko[e] = fromf<T>(rope_one<T>(k8[e], kp, c, sn));
            }
            stg16(qout + row * H + n0, as_u4<T>(qo));
            // (dd re-derived behind an opaque move: its loop-invariant kperm() address part was hoisted into a register the 128-register build then
            // spilled, and the reload inside the token loop came with an s_waitcnt vmcnt(0) -- round 5: no scratch)
            int dd_k = dd;
            asm volatile("" : "+v"(dd_k));
            stg16(kcache + ((size_t)b * d.heads + hh) * d.max_len * D + kperm(slot0 + t, dd_k, d.k_perm), as_u4<T>(ko));     // fragment order per 16 positions
        }
    }
}

void launch_rope_kv_prefill(int dtype, const LlamaDims& d, const void* qkv, const void* lora_bq, const void* lora_bv,
                            const void* cos_t, const void* sin_t, const int* pos_ids, void* qout, void* kcache,
                            void* vcache, int B, int T_, int slot0, hipStream_t s) {
    const int threads = ((d.hidden / 8 + 63) / 64) * 64;          // hidden <= 8192
    // tokens per workgroup: 1 while that still leaves the chip short of workgroups (a single prompt: 160); for batched prompts as many as make the
    // grid one round of the 512 workgroups the chip holds (two per CU): 32 x 160 tokens -> 10 (8 gave 640 workgroups: a second round a quarter full)
    int tpb = 1;
    if ((long)T_ * B >= 2048) { tpb = (int)(((long)T_ * B + 511) / 512); tpb = tpb < 4 ? 4 : (tpb > 16 ? 16 : tpb); }
    dim3 grid((T_ + tpb - 1) / tpb, B), block(threads);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((rope_kv_prefill_k<T>), grid, block, 0, s, d, (const T*)qkv, (const T*)lora_bq, (const T*)lora_bv,
                                                (const T*)cos_t, (const T*)sin_t, pos_ids, (T*)qout, (T*)kcache, (T*)vcache, B, T_, slot0, tpb));
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

constexpr int DA_WAVES_MID = 8;    // 9-16 rows (257-512 pairs): two 8-wave workgroups per CU, a 336-position register window each (the 4-wave form leaves a CU 8 waves).
                                   // Dedicated new-token wave, 3 K groups per wave, V behind the scores; measured against all 8 waves owning rows with a 256-position
                                   // window and V early (+0.05 ms per step at 12 / 16 rows) and the dedicated wave with a 224-position window and V early (+0.03)

template <typename T, int WAVES>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 16 ? 1 : 4)) void decode_attention_k(DecAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    // latency variant: wave 0 is dedicated to the new token (its operand loads are first in its queue), 15 waves own the cache
    if (WAVES == 16) decode_attention_body<T, WAVES, true, NoWait, true>(a, blockIdx.x, blockIdx.y, dsm);
    else if (WAVES == 8) decode_attention_body<T, WAVES, true, NoWait, false, 3>(a, blockIdx.x, blockIdx.y, dsm);
    else decode_attention_body<T, WAVES, false>(a, blockIdx.x, blockIdx.y, dsm);
}

void launch_decode_attention(int dtype, const DecAttnArgs& a, int B, hipStream_t s) {
    dim3 grid(a.d.heads, B);
    const char* tp_env = getenv("RDX_ATT_TP");                       // tests: 1 forces the throughput variant, 2 the 8-wave one (re-read per launch: the tests flip it between engines)
    const int force_tp = tp_env ? atoi(tp_env) : 0;
    static const bool att_mid = !(getenv("RDX_ATT_MID") && atoi(getenv("RDX_ATT_MID")) == 0);      // A/B switch, read once
    if (a.d.heads * B <= 256 && !force_tp) {
        const size_t smem = decode_attention_smem_floats(DA_WAVES, a.d.max_len) * sizeof(float);
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((decode_attention_k<T, DA_WAVES>), grid, dim3(DA_WAVES * 64), smem, s, a));
    } else if (force_tp == 2 || (a.d.heads * B <= 512 && !force_tp && att_mid)) {
        // 9-16 rows at 32 heads (round 5; RDX_ATT_MID=0: the 4-wave form, the A/B leg; RDX_ATT_TP=2: forced, tests): batch 12 3.162 -> 3.098 ms per step, 16: 3.222 -> 3.178
        const size_t smem = decode_attention_smem_floats(DA_WAVES_MID, a.d.max_len) * sizeof(float);
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((decode_attention_k<T, DA_WAVES_MID>), grid, dim3(DA_WAVES_MID * 64), smem, s, a));
    } else {
        const size_t smem = decode_attention_smem_floats(DA_WAVES_TP, a.d.max_len) * sizeof(float);
        RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((decode_attention_k<T, DA_WAVES_TP>), grid, dim3(DA_WAVES_TP * 64), smem, s, a));
    }
}

}  // namespace rdx

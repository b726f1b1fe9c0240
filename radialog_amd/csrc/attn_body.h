// Body of decode attention (one new token per batch row) as a device function: runs as its own kernel
// (attn.hip: decode_attention_k, 16 or 4 waves), as the producer role of the fused attention + o_proj launch
// (chain.hip: attn_oproj16_k, 16 waves). head_dim 128: one K/V cache row =
// 256 B.
//
// One workgroup owns one (batch row, head). The cache is small (L x 256 B for K and for V), so the work is latency- and
// issue-bound on ONE CU, and it is organised to keep both the dependent-load chain and the VALU instruction count short:
//   * every lane issues ALL its cache loads up front -- K as MFMA A-operand fragments (16 positions x 32 dims per
//     wave-wide 16-byte load, KG groups of 16 positions per wave), V as rows (VR = 4*KG rows per lane), the key-mask bytes
//     -- none of these addresses depends on `slot` (rows past it are simply unused);
//   * wave 0 alone handles the new token in registers: LoRA-B add (peft un-merged), rotate-half RoPE (partner d+-64 =
//     lane^8 via DPP), append to the cache, publish q (model dtype) through LDS. The cos/sin row comes from the per-row
//     copy the greedy-step kernel left behind, so there is no position -> table dependent load;
//   * scores = K . q on the matrix cores: A = 16 cached positions x 32 dims, B = q replicated over the 16 columns, so
//     after 4 MFMAs a lane holds the finished scores of 4 positions -- no unpacking, no cross-lane reduction;
//   * every wave recomputes the softmax statistics itself (DPP / readlane reductions, no block reduction);
//   * P is computed ONCE per cached row (lane `u` of each 16-lane row owns row u) and broadcast inside the row by DPP
//     row_share; P.V uses v_dot2 on (row u, row u+1) pairs, i.e. one permute + one dot per two MACs.
// Rounding points are the reference's (modeling_llama_imgemb.py:135-142,:198-234): q/k/v in T, T(T(q.k)/sqrt(d)),
// fp32 softmax, P rounded to T before PV, output rounded to T.
#pragma once
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "handoff.h"

namespace rdx {

template <typename T>
__device__ __forceinline__ float rope_one(float x, float partner_signed, float c, float s) {
    // (x * cos) + (rotate_half(x) * sin), every op rounded to the model dtype
    return rnd<T>(rnd<T>(x * c) + rnd<T>(partner_signed * s));
}

__host__ __device__ inline size_t decode_attention_smem_floats(int waves, int max_len) { return (size_t)waves * 128 + 128 + max_len; }

template <int N> __device__ __forceinline__ float row_share(float v) { return dpp_mov<0x150 + N>(v); }   // lane N of each 16-lane row

// a.x*b.x + a.y*b.y + c on packed model-dtype pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16)
template <typename T> __device__ __forceinline__ float dot2(unsigned a, unsigned b, float c);
template <> __device__ __forceinline__ float dot2<bf16>(unsigned a, unsigned b, float c) {
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2b, a), __builtin_bit_cast(v2b, b), c, false);
}
template <> __device__ __forceinline__ float dot2<f16>(unsigned a, unsigned b, float c) {
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h, a), __builtin_bit_cast(v2h, b), c, false);
}

template <typename T, int WAVES, bool DED = (WAVES == 8), typename WaitFn = NoWait, bool V_EARLY_ = !DED, int KG_ = 0, bool COH = false>
__device__ __forceinline__ void decode_attention_body(const DecAttnArgs& a, const int h, const int b, float* dsm,
                                                      WaitFn wait_inputs = WaitFn()) {
    typedef typename Vec8<T>::type V8;
    constexpr int D = 128;
    // DED (register-tight launches): wave 0 is dedicated to the new token and holds no cache rows, so its register-hungry
    // LoRA/RoPE block never overlaps a live K window; waves 1.. own the cache rows.
    constexpr int CW = DED ? WAVES - 1 : WAVES;  // waves that own cache rows
    constexpr int KG = KG_ ? KG_ : ((DED && WAVES == 8) ? 4 : 2);   // K groups (16 positions) per wave held in registers
    constexpr int VR = 4 * KG;                   // V rows per lane held in registers (same coverage: KG*CW*16 positions)
    constexpr int SPAN = CW * 4;                 // positions covered by one block-wide V row sweep
    constexpr bool V_EARLY = V_EARLY_;           // enough registers to have K and V in flight together
    // tail trips of contexts beyond the register window: K groups of 16 positions / V rows per lane and trip. Round 5 measured fatter trips in the
    // 4-wave build (it has 19 VGPRs to spare since the new token's registers are retired early): 3 groups, 8 rows or both are 1.0-1.3 us SLOWER at
    // every context from 96 to 448 (30.8 -> 31.9-32.1 us at 288) -- as with every earlier attempt to put more of this kernel's requests in flight
    constexpr int KT = 2, VT = 4;
    const LlamaDims& d = a.d;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const T* lbq = reinterpret_cast<const T*>(a.lbq);
    const T* lbv = reinterpret_cast<const T*>(a.lbv);
    const T* cos_t = reinterpret_cast<const T*>(a.cos_t);
    const T* sin_t = reinterpret_cast<const T*>(a.sin_t);

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* part = dsm;                                   // [WAVES][D]
    T* qT = reinterpret_cast<T*>(dsm + WAVES * D);       // [D] rotated query, model dtype (MFMA B operand)
    float* S = dsm + WAVES * D + D;                      // [max_len]

    const int jsub = lane >> 4, doct = lane & 15;        // V / new-token view: 16 lanes x 8 dims per cache row
    const int g = lane >> 4, r = lane & 15;              // K view (MFMA A fragment): position r of the group, dims 32*c + 8*g
    const int H = d.hidden;
    const T* x = qkv + (size_t)b * d.qkv_ld;
    T* kc = reinterpret_cast<T*>(a.kcache) + ((size_t)b * d.heads + h) * d.max_len * D;
    T* vc = reinterpret_cast<T*>(a.vcache) + ((size_t)b * d.heads + h) * d.max_len * D;
    const uint8_t* km = a.key_mask + (size_t)b * d.max_len;

    if (a.trace && h == 0 && b == 0 && tid == 0) a.trace[7] = (long long)__builtin_amdgcn_s_memrealtime();   // entry
    // ---- cache loads of this lane in flight first ------------------------------------------------------------------
    const int cw = DED ? w - 1 : w;              // cache-wave index (-1: the dedicated new-token wave)
    const bool owns_rows = !DED || w > 0;        // wave-uniform
    u4 kr[KG][4], vr[VR];
    unsigned kmw[KG];                            // key-mask bytes of positions gb + 4g .. +3
    // The lower half of the register window is loaded unconditionally at kernel entry; the upper half only once `slot`
    // is known and only by the waves whose rows exist (all of a head's cache traffic funnels through ONE CU's L1 at
    // 64 B/clk, so a fully loaded 480-position window costs ~2 us whatever the context length).
    constexpr int KG_LO = (KG + 1) / 2, VR_LO = 4 * KG_LO;
    auto load_k = [&](int u0, int u1, int limit) {
#pragma unroll
        for (int u = 0; u < KG; ++u) {
            const int gb = (u * CW + cw) * 16;
            if (u >= u0 && u < u1 && gb < limit) {
                const int j = min(gb + r, d.max_len - 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) kr[u][c] = ldg16(kc + kperm(j, c * 32 + g * 8, d.k_perm));
                kmw[u] = *reinterpret_cast<const unsigned*>(km + min(gb + 4 * g, d.max_len - 4));
            }
        }
    };
    auto load_v = [&](int u0, int u1, int limit) {
#pragma unroll
        for (int u = 0; u < VR; ++u) {
            // pair granularity: P.V consumes rows (u, u + 1) together whenever row u is inside the context, so the odd row of
            // a pair is loaded with its even partner (clamped address, P = 0) -- an unloaded register may hold a NaN pattern
            if (u >= u0 && u < u1 && (u & ~1) * SPAN + cw * 4 < limit) {
                const int j = min(u * SPAN + cw * 4 + jsub, d.max_len - 1);
                vr[u] = ldg16(vc + (size_t)j * D + doct * 8);
            }
        }
    };
    // operands of the new token (wave 0 only; wave-uniform branches). Stand-alone launches issue them FIRST, ahead of every
    // cache load of the workgroup; chained launches must wait for the producer and issue them after wait_inputs().
    constexpr bool HAS_WAIT = !__is_same(WaitFn, NoWait);
    const int n0 = h * D + doct * 8;
    u4 nq, nk_, nv, ncos, nsin, naq, nav, nbq0, nbq1, nbv0, nbv1;
    auto load_newtok = [&]() {
        // COH (chained launches): the qkv row was published write-through by other workgroups of THIS launch -> read it
        // with agent-scope loads (L1 bypass); everything else this kernel reads predates the launch
        auto ldx = [&](const T* p) -> u4 {
            if (!COH) return ldg16(p);
            const unsigned long long lo = ld8_agent(p), hi = ld8_agent(p + 4);
            return (u4){(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        };
        nq = ldx(x + n0); nk_ = ldx(x + H + n0); nv = ldx(x + 2 * H + n0);
        // cos/sin row of this token's position: from the per-row copy greedy_step_k left behind (no pos -> table
        // dependent load on the critical path), else from the tables
        const T* cs = reinterpret_cast<const T*>(a.cur_rope);
        const T* cp = cs ? cs + (size_t)b * 2 * D : cos_t + (size_t)a.pos[b] * D;
        const T* sp = cs ? cs + (size_t)b * 2 * D + D : sin_t + (size_t)a.pos[b] * D;
        ncos = ldg16(cp + doct * 8); nsin = ldg16(sp + doct * 8);
        if (d.lora_r == 8) {
            // The four jsub groups of the wave hold the same 8 dims; each takes two of the eight LoRA-B rows, so all
            // B loads are ONE round trip (4 x 16 B per lane), and the deltas come back through shuffles.
            const int e0 = 2 * jsub;
            naq = ldx(x + 3 * H); nav = ldx(x + 3 * H + 8);
            nbq0 = ldg16(lbq + (size_t)(n0 + e0) * 8); nbq1 = ldg16(lbq + (size_t)(n0 + e0 + 1) * 8);
            nbv0 = ldg16(lbv + (size_t)(n0 + e0) * 8); nbv1 = ldg16(lbv + (size_t)(n0 + e0 + 1) * 8);
        }
    };
    if (!HAS_WAIT && w == 0) load_newtok();
    if (owns_rows) {
        load_k(0, KG_LO, 0x7fffffff);            // unconditional (addresses clamped into the cache): these registers are
        if (V_EARLY) load_v(0, VR_LO, 0x7fffffff);   // always consumed, with P = 0 for rows past the context
    }
    const int slot = a.slot_b[b];
    const int nk = slot + 1;
    long long* trc = (a.trace && h == 0 && b == 0 && tid == 0) ? a.trace : nullptr;
#define ATT_T(i) do { if (trc) trc[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    ATT_T(0);
    wait_inputs();      // chained launches: the cache loads are already in flight; block until this step's qkv row is published
    ATT_T(1);

    // ---- new token (wave 0 only, in registers): LoRA add + RoPE for dims doct*8 .. +8 of this head -------------------
    float s_new = 0.f;                            // wave 0: q . k of the new position (fp32 sum, rounded where it is stored)
    if (w == 0) {
        float q8n[8], k8[8], v8[8];
        if (HAS_WAIT) load_newtok();
        const V8 qv = as_vec8<T>(nq), kv = as_vec8<T>(nk_), vv = as_vec8<T>(nv), cv = as_vec8<T>(ncos), sv_ = as_vec8<T>(nsin);
#pragma unroll
        for (int e = 0; e < 8; ++e) { q8n[e] = tof<T>(qv[e]); k8[e] = tof<T>(kv[e]); v8[e] = tof<T>(vv[e]); }
        if (d.lora_r == 8) {
            const V8 aq = as_vec8<T>(naq), av = as_vec8<T>(nav);
            const V8 bq0 = as_vec8<T>(nbq0), bq1 = as_vec8<T>(nbq1), bv0 = as_vec8<T>(nbv0), bv1 = as_vec8<T>(nbv1);
            float sq0 = 0.f, sq1 = 0.f, sv0 = 0.f, sv1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sq0 += tof<T>(bq0[i]) * tof<T>(aq[i]); sq1 += tof<T>(bq1[i]) * tof<T>(aq[i]);
                sv0 += tof<T>(bv0[i]) * tof<T>(av[i]); sv1 += tof<T>(bv1[i]) * tof<T>(av[i]);
            }
            sq0 = rnd<T>(rnd<T>(sq0) * d.lora_scale); sq1 = rnd<T>(rnd<T>(sq1) * d.lora_scale);   // lora_B(lora_A(x)) * scaling
            sv0 = rnd<T>(rnd<T>(sv0) * d.lora_scale); sv1 = rnd<T>(rnd<T>(sv1) * d.lora_scale);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int src = ((e >> 1) << 4) | doct;
                const float dq = __shfl((e & 1) ? sq1 : sq0, src, 64), dv = __shfl((e & 1) ? sv1 : sv0, src, 64);
                q8n[e] = rnd<T>(q8n[e] + dq);                                   // result += delta
                v8[e] = rnd<T>(v8[e] + dv);
            }
        }
        const bool lo = doct < 8;                                               // dims < 64: rotate_half gives -x[d+64]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float qo = dpp_mov<DPP_ROR8>(q8n[e]), ko = dpp_mov<DPP_ROR8>(k8[e]);     // lane ^ 8 inside the 16-lane row
            const float c = tof<T>(cv[e]), sn = tof<T>(sv_[e]);
            q8n[e] = rope_one<T>(q8n[e], lo ? -qo : qo, c, sn);
            k8[e] = rope_one<T>(k8[e], lo ? -ko : ko, c, sn);
        }
        if (jsub == 0) {                                                        // publish q, append k / v to the cache
            V8 qo, ko, vo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { qo[e] = fromf<T>(q8n[e]); ko[e] = fromf<T>(k8[e]); vo[e] = fromf<T>(v8[e]); }
            *reinterpret_cast<u4*>(qT + doct * 8) = as_u4<T>(qo);
            stg16(kc + kperm(slot, doct * 8, d.k_perm), as_u4<T>(ko));      // K rows live in the 16-position fragment order (rdx_common.h)
            stg16(vc + (size_t)slot * D + doct * 8, as_u4<T>(vo));
            // the new row's V (fp32) waits in wave 0's own slice of `part` until P.V: the same lanes read it back, and only then does wave 0
            // overwrite the slice with its partial output -- LDS serves a wave's operations in order, no barrier involved
#pragma unroll
            for (int e = 0; e < 8; ++e) part[doct * 8 + e] = v8[e];
        }
        // q . k of the new position, finished HERE: q8n / k8 / v8 (24 registers in every wave of the launch, used by wave 0 only) are dead
        // before the cache phases start -- they used to stay live through scores, softmax and P.V, and at the 128-register cap of the
        // 4-workgroups-per-CU and 16-wave builds the allocator spilled three to five dwords around them (round 5: no scratch)
        {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += q8n[e] * k8[e];
            s_new = row16_sum(acc);
        }
    }
    if (owns_rows) {                             // upper half of the window, now that the context length is known
        load_k(KG_LO, KG, slot);
        if (V_EARLY) load_v(VR_LO, VR, slot);
    }
    ATT_T(2);
    __syncthreads();
    ATT_T(3);

    // ---- scores: S[j] = T(T(k_j . q) / sqrt(D)) on the matrix cores ----------------------------------------------------------
    const float div = sqrtf((float)D);
    const int km_new = km[min(slot, d.max_len - 1)];   // mask byte of the new position: requested here, consumed behind the cached groups' scores
    u4 qb[4];                                    // B operand: q[32c + 8g .. +8], the same in all 16 columns
#pragma unroll
    for (int c = 0; c < 4; ++c) qb[c] = *reinterpret_cast<const u4*>(qT + c * 32 + g * 8);
    auto score_group = [&](const u4 (&kf)[4], unsigned maskw, int gb) {     // wave-uniform call: MFMA ignores EXEC
        v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = mfma16(as_vec8<T>(kf[c]), as_vec8<T>(qb[c]), acc);
        if (r == 0) {                            // lane (g, r): acc[i] = score of position gb + 4g + i (any column r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = gb + 4 * g + i;
                if (j < slot) S[j] = ((maskw >> (8 * i)) & 0xffu) ? rnd<T>(rnd<T>(acc[i]) / div) : -INFINITY;
            }
        }
    };
    if (owns_rows) {
#pragma unroll
        for (int u = 0; u < KG; ++u)
            if (u < KG_LO || (u * CW + cw) * 16 < slot) score_group(kr[u], kmw[u], (u * CW + cw) * 16);   // wave-uniform
        if (!V_EARLY) { load_v(0, VR_LO, 0x7fffffff); load_v(VR_LO, VR, slot); }     // K registers are free now
        // contexts beyond the register window: two groups per trip, all eight fragment loads issued before the first MFMA
        // (the register window is dead by now); a group past the context is loaded from a clamped row and stores nothing
        // (the second group of the last trip may lie past the context: wave-uniform skip -- it would store nothing, and its
        // loads are real HBM traffic: at batch 32 the clamped rows were a fifth of the kernel's bytes)
        for (int gi = KG * CW + cw; gi * 16 < slot; gi += KT * CW) {
            u4 kf[KT][4];
            unsigned mw[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                if (t > 0 && !((gi + t * CW) * 16 < slot)) break;       // wave-uniform: a group past the context would store nothing, and its loads are real HBM traffic
                const int gb = (gi + t * CW) * 16;
                const int j = min(gb + r, d.max_len - 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) kf[t][c] = ldg16(kc + kperm(j, c * 32 + g * 8, d.k_perm));
                mw[t] = *reinterpret_cast<const unsigned*>(km + min(gb + 4 * g, d.max_len - 4));
            }
#pragma unroll
            for (int t = 0; t < KT; ++t)
                if (t == 0 || (gi + t * CW) * 16 < slot) score_group(kf[t], mw[t], (gi + t * CW) * 16);      // positions >= slot are not stored
        }
    }
    if (w == 0 && lane == 0) S[slot] = km_new ? rnd<T>(rnd<T>(s_new) / div) : -INFINITY;      // the new position itself
    __syncthreads();
    ATT_T(4);

    // ---- softmax statistics (fp32), recomputed by every wave -------------------------------------------------------
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, S[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) sum += expf(S[j] - mx);
    sum = wave_sum(sum);
    ATT_T(5);

    // ---- O = P V with P rounded to T ---------------------------------------------------------------------------------
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = 0.f;
    if (owns_rows) {
        // lane `u` of each 16-lane row computes P of cached row u (once), the row fetches it by DPP row_share
        float myp = 0.f;
        {
            const int j = doct * SPAN + cw * 4 + jsub;
            if (doct < VR && j < slot) myp = rnd<T>(expf(S[j] - mx) / sum);
        }
        float pu[16];
        pu[0] = row_share<0>(myp);   pu[1] = row_share<1>(myp);   pu[2] = row_share<2>(myp);   pu[3] = row_share<3>(myp);
        pu[4] = row_share<4>(myp);   pu[5] = row_share<5>(myp);   pu[6] = row_share<6>(myp);   pu[7] = row_share<7>(myp);
        pu[8] = row_share<8>(myp);   pu[9] = row_share<9>(myp);   pu[10] = row_share<10>(myp); pu[11] = row_share<11>(myp);
        pu[12] = row_share<12>(myp); pu[13] = row_share<13>(myp); pu[14] = row_share<14>(myp); pu[15] = row_share<15>(myp);
#pragma unroll
        for (int u = 0; u < VR; u += 2) {
            if (u >= VR_LO && u * SPAN + cw * 4 >= slot) continue;               // wave-uniform: rows were never loaded
            // rows u, u+1 of this lane: (v_u[d], v_u+1[d]) . (p_u, p_u+1) for its 8 dims; rows >= slot have p = 0
            // (the cache is zero-initialised, so unused rows hold finite values)
            const unsigned pp = (unsigned)bits16<T>(fromf<T>(pu[u])) | ((unsigned)bits16<T>(fromf<T>(pu[u + 1])) << 16);
            const unsigned a0[4] = {vr[u].x, vr[u].y, vr[u].z, vr[u].w};
            const unsigned a1[4] = {vr[u + 1].x, vr[u + 1].y, vr[u + 1].z, vr[u + 1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = __builtin_amdgcn_perm(a1[e], a0[e], 0x05040100u);    // (v_u[2e],   v_u+1[2e])
                const unsigned hi = __builtin_amdgcn_perm(a1[e], a0[e], 0x07060302u);    // (v_u[2e+1], v_u+1[2e+1])
                o8[2 * e] = dot2<T>(lo, pp, o8[2 * e]);
                o8[2 * e + 1] = dot2<T>(hi, pp, o8[2 * e + 1]);
            }
        }
        // contexts beyond the register window: four rows per trip, loads first (clamped rows get P = 0)
        for (int j0 = VR * SPAN; j0 < slot; j0 += VT * SPAN) {
            u4 vt[VT];
            float pt[VT];
#pragma unroll
            for (int t = 0; t < VT; ++t) {
                const int j = j0 + t * SPAN + cw * 4 + jsub;
                // rows of a whole wave past the context (wave-uniform test) are not fetched: zero row, P = 0
                vt[t] = (j0 + t * SPAN + cw * 4 < slot) ? ldg16(vc + (size_t)min(j, d.max_len - 1) * D + doct * 8) : (u4){0u, 0u, 0u, 0u};
                pt[t] = j < slot ? rnd<T>(expf(S[j] - mx) / sum) : 0.f;
            }
#pragma unroll
            for (int t = 0; t < VT; t += 2) {
                const unsigned pp = (unsigned)bits16<T>(fromf<T>(pt[t])) | ((unsigned)bits16<T>(fromf<T>(pt[t + 1])) << 16);
                const unsigned a0[4] = {vt[t].x, vt[t].y, vt[t].z, vt[t].w};
                const unsigned a1[4] = {vt[t + 1].x, vt[t + 1].y, vt[t + 1].z, vt[t + 1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned lo = __builtin_amdgcn_perm(a1[e], a0[e], 0x05040100u);
                    const unsigned hi = __builtin_amdgcn_perm(a1[e], a0[e], 0x07060302u);
                    o8[2 * e] = dot2<T>(lo, pp, o8[2 * e]);
                    o8[2 * e + 1] = dot2<T>(hi, pp, o8[2 * e + 1]);
                }
            }
        }
    }
    if (w == 0 && jsub == 0) {
        const float p = rnd<T>(expf(S[slot] - mx) / sum);
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] += p * part[doct * 8 + e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = xor32_sum(xor16_sum(o8[e]));
    if (jsub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[w * D + doct * 8 + e] = o8[e];
    }
    __syncthreads();
    ATT_T(6);
    if (!COH) {
        if (tid < D) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) v += part[i * D + tid];
            const int k = h * D + tid;
            // out_packed: the 32-row fragment-packed block the K-split o_proj reads (xsplit32_k): [k / 32][b / 16][(k % 32) / 8][b % 16][8]
            // (out_packed 2: the fp8 consumer's 64-deep order, fragment 2 (k / 64) + (k % 16) / 8, g = (k % 64) / 16)
            const int pf = a.out_packed == 2 ? 2 * (k >> 6) + ((k & 15) >> 3) : (k >> 5), pg = a.out_packed == 2 ? ((k & 63) >> 4) : ((k & 31) >> 3);
            // (out_packed 2 beyond 32 rows: one such 32-row block per 32 rows, at a block stride of 32 H elements)
            const size_t o = a.out_packed == 2 ? (size_t)(b >> 5) * (32 * H) + (((size_t)((pf * 2 + ((b >> 4) & 1)) * 64 + pg * 16 + (b & 15)) << 3) + (k & 7))
                           : a.out_packed ? ((size_t)((pf * a.out_mt + (b >> 4)) * 64 + pg * 16 + (b & 15)) << 3) + (k & 7) : (size_t)b * H + k;
            reinterpret_cast<T*>(a.out)[o] = fromf<T>(v);
        }
    } else if (tid < D / 4) {                                  // write-through 8-byte stores (4 dims per lane)
        unsigned long long pk = 0ull;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) v += part[i * D + tid * 4 + e];
            pk |= (unsigned long long)bits16<T>(fromf<T>(v)) << (16 * e);
        }
        st8_agent(reinterpret_cast<T*>(a.out) + (size_t)b * H + h * D + tid * 4, pk);
    }
#undef ATT_T
}

}  // namespace rdx

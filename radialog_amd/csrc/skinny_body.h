// Body of the skinny (M <= 32) weight-streaming GEMM, as a device function so that it can run either as its own kernel
// (gemm.hip: skinny_gemm_k) or as the consumer role of a fused launch (chain.hip: attention + o_proj, where the tile's
// weights are already in flight while the producer workgroups are still computing the activations).
//
// One workgroup = one 16-column output tile; its WAVES waves split K. Each wave streams its K slice of the packed tile
// (1 KiB per wave-instruction, non-temporal) in register batches of U blocks, two batches deep; both are issued BEFORE
// `wait_inputs()` and the prologue, so HBM latency overlaps them. All weight loads are unconditional (addresses clamped
// inside the slice) so the compiler can hoist a whole batch.
// XLDS = true : M*K activations fit in LDS -> the workgroup normalises (optional fused RMSNorm) and stages x-hat once,
//               the main loop takes its MFMA B operand from LDS (no per-wave redundant norm math, few registers).
// XLDS = false: activations streamed from global/L2 in fragment order (already normalised by rmsnorm_k if needed).
#pragma once
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "handoff.h"

namespace rdx {

// shared epilogue math (rounding points follow the reference's fp16/bf16 op sequence, see oracle/ref_cpu.py)
template <typename T> __device__ __forceinline__ float swiglu(float gate_acc, float up_acc) {
    const float gt = rnd<T>(gate_acc), up = rnd<T>(up_acc);   // gate_proj(x), up_proj(x) as model-dtype tensors
    const float s = rnd<T>(silu(gt));                          // act_fn output rounded
    return s * up;                                             // product rounded by the caller's store
}


template <typename T, int MT, int EPI, bool NORM, int WAVES, bool XLDS, typename WaitFn, bool COHX = false, bool W8 = false>
__device__ __forceinline__ void skinny_tile(const GemmArgs& a, const int tile, const int ntiles, unsigned char* dyn_smem,
                                            WaitFn wait_inputs) {
    typedef typename Vec8<T>::type V8;
    // W8: fp8 (e4m3) weights with one fp32 scale per output row. A chunk is then 64 k-values (16 bytes per lane, two MFMAs:
    // lane (g, r) holds W[16*tile + r][64*c + 16*g .. +16]); the bytes are expanded to the model dtype in registers (exact)
    // and the scale is applied to the fp32 sums in the epilogue. LDS-staged activations only.
    static_assert(!W8 || (XLDS && MT == 1), "fp8 weights: LDS-staged activations, one M tile");
    constexpr int U = (XLDS && WAVES >= 8) ? 8 : 4;    // 4-wave workgroups stay <= 64 VGPRs: 8 workgroups per CU
    constexpr int NTHR = WAVES * 64;
    __shared__ __attribute__((aligned(16))) float red[WAVES][MT][256];   // [wave][mt][m_local*16 + n_local]
    __shared__ float ssq[WAVES][16];
    __shared__ float rstd_s[16];
    T* xs = reinterpret_cast<T*>(dyn_smem);                              // XLDS: [M][K] x-hat

    // wave id made provably wave-uniform (SGPR): K-slice bounds become scalar, and no MFMA ends up under an
    // EXEC-masked per-lane branch (MFMA ignores EXEC -- a masked-off MFMA would still accumulate)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int K = a.K, KC = W8 ? (K >> 6) : (K >> 5);
    const int c0 = (KC * w) / WAVES, c1 = (KC * (w + 1)) / WAVES;
    const T* X = reinterpret_cast<const T*>(a.X);
    const u4* wbase = reinterpret_cast<const u4*>(W8 ? a.W8 : a.W) + (size_t)tile * KC * 64 + lane;
    const int clast = min(max(c1 - 1, c0), KC - 1);

    long long* trc = (a.trace && threadIdx.x == 0) ? a.trace + (size_t)tile * 8 : nullptr;
#define SK_T(i) do { if (trc) trc[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    SK_T(0);
    if (trc) {       // where this workgroup runs: HW_ID (wave/simd/cu/sh/se fields) and the XCC (XCD) id
        trc[6] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID (id 4), bits 0..31
        trc[7] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID (id 20)
    }
    // two weight batches in flight before anything else (HBM latency overlaps the wait and the prologue)
    // (unconditional, addresses clamped into the slice: a short slice re-reads its last KiB)
    u4 ring[2 * U];
#pragma unroll
    for (int u = 0; u < 2 * U; ++u) ring[u] = ldg16_nt(wbase + (size_t)min(c0 + u, clast) * 64);
    u4* const wv = ring;            // the two-batch view of the direct-activation path below
    u4* const wn = ring + U;

    SK_T(1);
    wait_inputs();          // fused launches: block until the producer workgroups have published X (and resid)

    // COHX (fused launches with the fence-free hand-off): X was published write-through by other workgroups of THIS
    // launch -> agent-scope loads (L1 bypass); resid / weights predate the launch
    auto ldxa = [&](const T* p) -> u4 {
        if (!COHX) return ldg16(p);
        const unsigned long long lo = ld8_agent(p), hi = ld8_agent(p + 4);
        return (u4){(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    };
    if (XLDS) {
        const int K8 = K >> 3;
        if (NORM) {
            // LlamaRMSNorm (:85-93): fp32 mean of squares per row
            for (int m = 0; m < a.M; ++m) {
                float ss = 0.f;
                for (int k8 = threadIdx.x; k8 < K8; k8 += NTHR) {
                    V8 xv = as_vec8<T>(ldxa(X + (size_t)m * a.ldx + (size_t)k8 * 8));
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = tof<T>(xv[j]); ss += f * f; }
                }
                ss = wave_sum(ss);
                if (lane == 0) ssq[w][m] = ss;
            }
            __syncthreads();
            if (threadIdx.x < a.M) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) t += ssq[i][threadIdx.x];
                rstd_s[threadIdx.x] = rsqrtf(t / (float)K + a.eps);
            }
            __syncthreads();
        }
        const T* NW = reinterpret_cast<const T*>(a.norm_w);
        const int total8 = a.M * K8;
        for (int i = threadIdx.x; i < total8; i += NTHR) {
            const int m = i / K8, k8 = i - m * K8;
            V8 xv = as_vec8<T>(ldxa(X + (size_t)m * a.ldx + (size_t)k8 * 8));
            if (NORM) {
                const float rs = rstd_s[m];
                V8 nw = as_vec8<T>(ldg16(NW + (size_t)k8 * 8));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float h = rnd<T>(tof<T>(xv[j]) * rs);      // (x * rsqrt(var+eps)).to(dtype)
                    xv[j] = fromf<T>(tof<T>(nw[j]) * h);             // weight * hidden  (dtype mult)
                }
            }
            *reinterpret_cast<u4*>(xs + (size_t)m * K + (size_t)k8 * 8) = as_u4<T>(xv);
        }
        __syncthreads();
    }

    SK_T(2);                // activations staged
    const T* xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + r;
        xok[mt] = m < a.M;
        xrow[mt] = XLDS ? (xs + (size_t)(xok[mt] ? m : 0) * K + g * (W8 ? 16 : 8)) : (X + (size_t)(xok[mt] ? m : 0) * a.ldx + g * 8);
    }

    v4f acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};

    // activations of the non-LDS path are software-pipelined one batch ahead as well (L2 latency, 2x the weight bytes at M = 32)
    u4 xr[U][MT], xn[U][MT];
    auto load_x = [&](u4 (&dst)[U][MT], int cb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = min(cb + u, clast);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                dst[u][mt] = xok[mt] ? ldxa(xrow[mt] + (size_t)c * 32) : (u4){0u, 0u, 0u, 0u};
        }
    };
    if (!XLDS) load_x(xr, c0);

    if (XLDS) {
        // LDS-staged activations (every decode GEMV): a ring of 2 U weight fragments, each refilled right after the MFMAs that consumed it, the pair
        // pinned by a scheduling barrier -- the waits are vmcnt(2 U - 1), 2 U - 1 loads per wave stay in flight. (As two batches with a register
        // copy "wv = wn" between them the loop waited vmcnt(0) on every trip.) MFMA columns of rows >= M read row 0 again (LDS broadcast, never
        // stored): a per-lane select around the LDS read would put an EXEC-masked branch into the loop.
        auto consume1 = [&](const u4& wreg, int c) {
            if (W8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u4 wd = dequant8<T>(h ? wreg.z : wreg.x, h ? wreg.w : wreg.y);
                    const u4 xv = *reinterpret_cast<const u4*>(xrow[0] + (size_t)c * 64 + h * 8);
                    acc[0] = mfma16(as_vec8<T>(wd), as_vec8<T>(xv), acc[0]);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u4 xv = *reinterpret_cast<const u4*>(xrow[mt] + (size_t)c * 32);
                    acc[mt] = mfma16(as_vec8<T>(wreg), as_vec8<T>(xv), acc[mt]);
                }
            }
        };
        int cb = c0;
        for (; cb + 2 * U < c1; cb += 2 * U) {
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) {
                consume1(ring[u], cb + u);
                ring[u] = ldg16_nt(wbase + (size_t)min(cb + 2 * U + u, clast) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * U; ++u)
            if (cb + u < c1) consume1(ring[u], cb + u);
    } else
    for (int cb = c0; cb < c1; cb += U) {
        if (!XLDS && cb + U < c1) load_x(xn, cb + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (W8) {
                if (cb + u < c1) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u4 wd = dequant8<T>(h ? wv[u].z : wv[u].x, h ? wv[u].w : wv[u].y);
                        const u4 xv = xok[0] ? *reinterpret_cast<const u4*>(xrow[0] + (size_t)(cb + u) * 64 + h * 8) : (u4){0u, 0u, 0u, 0u};
                        acc[0] = mfma16(as_vec8<T>(wd), as_vec8<T>(xv), acc[0]);
                    }
                }
            } else if (cb + u < c1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    u4 xv;
                    if (XLDS) xv = xok[mt] ? *reinterpret_cast<const u4*>(xrow[mt] + (size_t)(cb + u) * 32) : (u4){0u, 0u, 0u, 0u};
                    else xv = xr[u][mt];
                    acc[mt] = mfma16(as_vec8<T>(wv[u]), as_vec8<T>(xv), acc[mt]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) wv[u] = wn[u];
        if (!XLDS && cb + U < c1) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xr[u][mt] = xn[u][mt];
        }
        if (cb + 2 * U < c1) {
#pragma unroll
            for (int u = 0; u < U; ++u) wn[u] = ldg16_nt(wbase + (size_t)min(cb + 2 * U + u, clast) * 64);
        }
    }
    SK_T(3);                // K loop done (wave 0)
    // D[i = n_local = g*4+reg][j = m_local = r]  ->  red[w][mt][m_local*16 + n_local]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<float4*>(&red[w][mt][r * 16 + g * 4]) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    __syncthreads();
    SK_T(4);                // all waves done

    const int t = threadIdx.x;
    constexpr int NOUT = MT * 256;
    for (int o = t; o < NOUT; o += NTHR) {
        const int mt = o >> 8, idx = o & 255, m_local = idx >> 4, n_local = idx & 15;
        const int m = mt * 16 + m_local;
        const int n = tile * 16 + n_local;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) v += red[i][mt][idx];
        if (W8) v *= a.wscale[n];                                      // per-row dequantisation scale (padded rows: 1)
        if (a.bias && n < a.N) v += a.bias[n];
        T* out = reinterpret_cast<T*>(a.out);
        if (a.out_step && out) out += (size_t)(*a.out_step) * a.out_step_stride;
        const bool ok = (m < a.M) && (n < a.N);

        if (EPI == EPI_NONE) {
            if (ok) out[(size_t)m * a.ldo + n] = fromf<T>(v);
        } else if (EPI == EPI_RELU) {
            if (ok) out[(size_t)m * a.ldo + n] = fromf<T>(fmaxf(v, 0.f));
        } else if (EPI == EPI_GELU) {
            if (ok) out[(size_t)m * a.ldo + n] = fromf<T>(gelu_erf(v));
        } else if (EPI == EPI_RESID) {
            if (ok) {
                const float rsd = tof<T>(reinterpret_cast<const T*>(a.resid)[(size_t)m * a.ldr + n]);
                out[(size_t)m * a.ldo + n] = fromf<T>(rsd + rnd<T>(v));
            }
        } else if (EPI == EPI_SILU_MUL) {
            // rows of a tile: 0..7 = gate_proj rows 8t..8t+7, 8..15 = up_proj rows 8t..8t+7
            float u = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) u += red[i][mt][(idx + 8) & 255];
            if (W8) u *= a.wscale[tile * 16 + ((n_local + 8) & 15)];
            if (n_local < 8 && ok) out[(size_t)m * a.ldo + tile * 8 + n_local] = fromf<T>(swiglu<T>(v, u));
        } else if (EPI == EPI_LOGITS) {
            float lv = rnd<T>(v);
            int li = n;
            const bool valid = n < a.n_valid;
            if (valid && m < a.M && out) out[(size_t)m * a.ldo + n] = fromf<T>(lv);
            if (!valid) { lv = -INFINITY; li = 0x7fffffff; }
            // argmax over the tile's 16 columns (16 consecutive lanes share m); ties -> lowest index (torch.argmax)
#pragma unroll
            for (int sh = 8; sh > 0; sh >>= 1) {
                const float ov = __shfl_xor(lv, sh, 64);
                const int oi = __shfl_xor(li, sh, 64);
                if (ov > lv || (ov == lv && oi < li)) { lv = ov; li = oi; }
            }
            if (n_local == 0 && m < a.M) {
                a.part_val[(size_t)m * ntiles + tile] = lv;
                a.part_idx[(size_t)m * ntiles + tile] = li;
            }
        }
    }
    SK_T(5);
#undef SK_T
}

}  // namespace rdx

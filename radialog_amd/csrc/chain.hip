// Chained decode launches of the batch-1/2 step (gfx950): units of consecutive decode layers run as ROLES of one launch, the dependency
// between them is a counter hand-off (handoff.h) instead of a kernel boundary -- a consumer workgroup first puts two register batches
// of ITS weight tile in flight, then waits until the producer role has published the activations, so HBM stays busy across the seam.
//
//   decode_chain_k   down_proj(l) (+ residual)  ->  RMSNorm + QKV(l + 1): one launch per layer, one 1024-thread workgroup per CU
//                    (two resident 16-wave workgroups per CU run every unit ~25 % slower, so the launch reserves > half of the LDS);
//   attn_oproj16_k   decode attention (attn_body.h) -> o_proj (+ residual): 32 attention workgroups + 128 two-tile o_proj workgroups
//                    whose whole K slice sits in registers while attention runs.
// Payloads are written write-through (8-byte agent-scope stores) and read with agent-scope loads: no cache fences. Arithmetic,
// rounding points and the fixed-order LDS reduction are those of skinny_body.h (same oracle parity).
// Liveness: a workgroup only waits on counters fed by lower-indexed workgroups; spins are bounded (handoff.h).
// (Round 3: the all-roles chained kernel RDX_MEGA, the gate/up -> down -> QKV chain RDX_CHAIN=1 and the 8-wave fused attention of
// fused.hip were measured slower in rounds 1-2 and have been removed; DESIGN.md 4 keeps the measurements.)
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "attn_body.h"
#include "handoff.h"
#include "skinny_body.h"   // swiglu()

namespace rdx {

constexpr int CH_WAVES = 16;
constexpr int CH_THREADS = CH_WAVES * 64;
constexpr int CH_MAXM = 2;                       // batch rows supported (LDS staging of [M][inter] activations)

struct ChainGemm {                                // one weight-streaming unit
    const void* X; int ldx;
    const void* W;
    const void* resid; int ldr;
    void* out; int ldo;
    int M, N, K;
    const void* norm_w; float eps;
    const void* W8; const float* wscale;         // fp8 weights (64-deep fragment order) + per-row scale, see skinny_body.h
};

template <typename T, int EPI, bool NORM, int SUB, int XL, typename WaitFn, int U = 4, bool RESID_EARLY = false, bool W8 = false>
__device__ __forceinline__ void chain_tile(const ChainGemm& a, const int wg, const int ntiles, unsigned char* smem, WaitFn wait_inputs) {
    typedef typename Vec8<T>::type V8;
    constexpr int WPS = CH_WAVES / SUB;           // waves per tile
    // U chunks per register batch, two batches in flight: U = 4 keeps every role of the chained kernel <= 64 VGPRs
    float* red = reinterpret_cast<float*>(smem);                        // [CH_WAVES][256]
    float* ssq = red + CH_WAVES * 256;                                  // [CH_WAVES][CH_MAXM]
    float* rstd_s = ssq + CH_WAVES * CH_MAXM;                           // [CH_MAXM] (+pad)
    T* xs = reinterpret_cast<T*>(rstd_s + 16);                          // [M][K] x-hat

    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = wa / WPS, w = wa - sub * WPS;
    const int tile = wg * SUB + sub;
    const int tile_c = min(tile, ntiles - 1);     // ragged last workgroup: duplicate loads, masked stores
    const int r = lane & 15, g = lane >> 4;
    const int K = a.K, KC = W8 ? (K >> 6) : (K >> 5);                   // W8: a chunk is 64 k-values (16 fp8 bytes per lane, two MFMAs)
    const int c0 = (KC * w) / WPS, c1 = (KC * (w + 1)) / WPS;
    const T* X = reinterpret_cast<const T*>(a.X);
    const u4* wbase = reinterpret_cast<const u4*>(W8 ? a.W8 : a.W) + (size_t)tile_c * KC * 64 + lane;
    const int clast = min(max(c1 - 1, c0), KC - 1);

    // two register batches (A, B) in flight before anything else; the main loop ping-pongs between them (no register
    // rotation: a rotated pair makes the compiler wait for the batch it has just issued)
    u4 wa_[U], wb_[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wa_[u] = ldg16_nt(wbase + (size_t)min(c0 + u, clast) * 64);
#pragma unroll
    for (int u = 0; u < U; ++u) wb_[u] = ldg16_nt(wbase + (size_t)min(c0 + U + u, clast) * 64);

    // RESID_EARLY: the residual predates the launch (fused attention + o_proj) -> fetch it now, off the critical tail
    constexpr int QPT_E = EPI == EPI_SILU_MUL ? 2 : 4;
    unsigned long long rs8_early = 0ull;
    if (RESID_EARLY && EPI == EPI_RESID && (int)threadIdx.x < SUB * a.M * QPT_E) {
        const int so = threadIdx.x / (a.M * QPT_E), rem = threadIdx.x - so * (a.M * QPT_E);
        const int m = rem / QPT_E, q = rem - m * QPT_E;
        const int n = (wg * SUB + so) * 16 + q * 4;
        if (wg * SUB + so < ntiles && n < a.N) rs8_early = ld8_agent(reinterpret_cast<const T*>(a.resid) + (size_t)m * a.ldr + n);
    }

    wait_inputs();

    // activations: published write-through by other workgroups of this launch -> agent-scope 8-byte loads (L1 bypass),
    // each chunk loaded ONCE and kept in registers across the RMSNorm statistics
    // XL = chunks of 4 elements per thread: M*K/4 <= XL*1024 (checked by chain_supported)
    const int K4 = K >> 2, total4 = a.M * K4;
    unsigned long long xr[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int c = threadIdx.x + i * CH_THREADS;
        xr[i] = 0ull;
        if (c < total4) { const int m = c / K4, k4 = c - m * K4; xr[i] = ld8_agent(X + (size_t)m * a.ldx + (size_t)k4 * 4); }
    }
    if (NORM) {
        float ss[CH_MAXM];
#pragma unroll
        for (int m = 0; m < CH_MAXM; ++m) ss[m] = 0.f;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int c = threadIdx.x + i * CH_THREADS;
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float f = tof<T>(from_bits16<T>((unsigned short)(xr[i] >> (16 * j)))); t += f * f; }
            const int m = c < total4 ? c / K4 : 0;                       // padding chunks are zero
#pragma unroll
            for (int mm = 0; mm < CH_MAXM; ++mm) ss[mm] += (m == mm) ? t : 0.f;
        }
#pragma unroll
        for (int m = 0; m < CH_MAXM; ++m) {
            const float t = wave_sum(ss[m]);
            if (lane == 0) ssq[wa * CH_MAXM + m] = t;
        }
        __syncthreads();
        if (threadIdx.x < CH_MAXM) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < CH_WAVES; ++i) t += ssq[i * CH_MAXM + threadIdx.x];
            rstd_s[threadIdx.x] = rsqrtf(t / (float)K + a.eps);
        }
        __syncthreads();
    }
    {
        const T* NW = reinterpret_cast<const T*>(a.norm_w);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int c = threadIdx.x + i * CH_THREADS;
            if (c < total4) {
                const int m = c / K4, k4 = c - m * K4;
                unsigned long long v = xr[i];
                if (NORM) {
                    const float rs = rstd_s[m];
                    const unsigned long long nw = *reinterpret_cast<const unsigned long long*>(NW + (size_t)k4 * 4);
                    v = 0ull;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float h = rnd<T>(tof<T>(from_bits16<T>((unsigned short)(xr[i] >> (16 * j)))) * rs);   // (x * rsqrt(var+eps)).to(dtype)
                        const float wv_ = tof<T>(from_bits16<T>((unsigned short)(nw >> (16 * j))));
                        v |= (unsigned long long)bits16<T>(fromf<T>(wv_ * h)) << (16 * j);                         // weight * hidden
                    }
                }
                *reinterpret_cast<unsigned long long*>(xs + (size_t)m * K + (size_t)k4 * 4) = v;
            }
        }
        __syncthreads();
    }

    // MFMA columns m >= M read row 0 again (an LDS broadcast) instead of zeros: their results are never stored, and a per-lane select around the
    // LDS read put an EXEC-masked branch inside the main loop -- the loop fell apart into basic blocks, the refill of a batch was issued before
    // the batch's last MFMA into a spare register, and the copy back waited vmcnt(0) on every trip
    const T* xrow = xs + (size_t)(r < a.M ? r : 0) * K + g * (W8 ? 16 : 8);
    v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
    // Main loop: loads are UNCONDITIONAL (addresses clamped into the slice) and no MFMA is guarded, so the compiler can
    // wait with counted vmcnt (one batch stays in flight); a conditional load anywhere in the loop makes it fall back
    // to vmcnt(0). At most one batch per wave is fetched in vain (same 1 KiB line as the slice's last chunk).
    auto consume = [&](const u4 (&wreg)[U], int cb, bool guard) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!guard || cb + u < c1) {
                if (W8) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u4 wd = dequant8<T>(h ? wreg[u].z : wreg[u].x, h ? wreg[u].w : wreg[u].y);
                        const u4 xv = *reinterpret_cast<const u4*>(xrow + (size_t)(cb + u) * 64 + h * 8);
                        acc = mfma16(as_vec8<T>(wd), as_vec8<T>(xv), acc);
                    }
                } else {
                    const u4 xv = *reinterpret_cast<const u4*>(xrow + (size_t)(cb + u) * 32);
                    acc = mfma16(as_vec8<T>(wreg[u]), as_vec8<T>(xv), acc);
                }
            }
        }
    };
    // Main loop: a ring of 2 U fragments, every fragment refilled right after the MFMA that consumed it, the pair pinned by a scheduling barrier
    // (the xstat32_k idiom): the waits are vmcnt(2 U - 1) and 2 U - 1 loads per wave stay in flight. As two batches ("consume U, refill U") the
    // compiler issued the refills early into spare registers and copied them back at the loop end behind vmcnt(1..7) / vmcnt(0): one load in
    // flight per wave at the end of every trip.
    auto consume1 = [&](const u4& wv, int c) {
        if (W8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u4 wd = dequant8<T>(h ? wv.z : wv.x, h ? wv.w : wv.y);
                const u4 xv = *reinterpret_cast<const u4*>(xrow + (size_t)c * 64 + h * 8);
                acc = mfma16(as_vec8<T>(wd), as_vec8<T>(xv), acc);
            }
        } else {
            const u4 xv = *reinterpret_cast<const u4*>(xrow + (size_t)c * 32);
            acc = mfma16(as_vec8<T>(wv), as_vec8<T>(xv), acc);
        }
    };
    int cb = c0;
    for (; cb + 2 * U < c1; cb += 2 * U) {
#pragma unroll
        for (int u = 0; u < 2 * U; ++u) {
            u4& wr = u < U ? wa_[u] : wb_[u - U];
            consume1(wr, cb + u);
            wr = ldg16_nt(wbase + (size_t)min(cb + 2 * U + u, clast) * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    consume(wa_, cb, true);
    consume(wb_, cb + U, true);
    // D[n_local = g*4+reg][m_local = r] -> red[wave][m_local*16 + n_local]
    *reinterpret_cast<float4*>(&red[wa * 256 + r * 16 + g * 4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();

    // epilogue: one thread per 4 consecutive output columns (one write-through 8-byte store each)
    T* out = reinterpret_cast<T*>(a.out);
    constexpr int QPT = EPI == EPI_SILU_MUL ? 2 : 4;                     // 4-column groups per tile row that are stored
    const int items = SUB * a.M * QPT;
    if ((int)threadIdx.x < items) {
        const int so = threadIdx.x / (a.M * QPT), rem = threadIdx.x - so * (a.M * QPT);
        const int m = rem / QPT, q = rem - m * QPT;
        const int t_o = wg * SUB + so;
        const int n = t_o * 16 + q * 4;
        const bool ok = (t_o < ntiles) && (n < a.N);                     // N % 4 == 0: a group is all-in or all-out
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = 0.f;
#pragma unroll
            for (int i = 0; i < WPS; ++i) v[j] += red[(so * WPS + i) * 256 + m * 16 + q * 4 + j];
            if (W8) v[j] *= a.wscale[t_o * 16 + q * 4 + j];
        }
        unsigned long long pk = 0ull;
        if (EPI == EPI_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pk |= (unsigned long long)bits16<T>(fromf<T>(v[j])) << (16 * j);
            if (ok) st8_agent(out + (size_t)m * a.ldo + n, pk);
        } else if (EPI == EPI_RESID) {
            if (ok) {
                const unsigned long long rs8 = RESID_EARLY ? rs8_early : ld8_agent(reinterpret_cast<const T*>(a.resid) + (size_t)m * a.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rsd = tof<T>(from_bits16<T>((unsigned short)(rs8 >> (16 * j))));
                    pk |= (unsigned long long)bits16<T>(fromf<T>(rsd + rnd<T>(v[j]))) << (16 * j);
                }
                st8_agent(out + (size_t)m * a.ldo + n, pk);
            }
        } else if (EPI == EPI_SILU_MUL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = 0.f;
#pragma unroll
                for (int i = 0; i < WPS; ++i) u += red[(so * WPS + i) * 256 + m * 16 + 8 + q * 4 + j];
                if (W8) u *= a.wscale[t_o * 16 + 8 + q * 4 + j];
                pk |= (unsigned long long)bits16<T>(fromf<T>(swiglu<T>(v[j], u))) << (16 * j);
            }
            if (ok) st8_agent(out + (size_t)m * a.ldo + t_o * 8 + q * 4, pk);
        }
    }
}

// down_proj(l) then RMSNorm + QKV(l + 1): blocks [0, nwg_down) are down_proj tiles (one tile per workgroup, 16 waves split K; their
// input predates the launch), blocks [nwg_down, +nwg_qkv) are QKV workgroups of the NEXT layer (4 tiles each, 4 waves per tile, the
// normalised row staged once per workgroup) that wait for all down_proj tiles on the layer's sharded counter
template <typename T, bool W8>
__global__ __launch_bounds__(CH_THREADS, 4) void decode_chain_k(ChainArgs ca) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    const ChainLayer& L = ca.layers[ca.layer];
    int* ctr = ca.ctr + (size_t)ca.layer * HO_CTR_INTS;
    const int H = ca.hidden, B = ca.B;
    if ((int)blockIdx.x < ca.nwg_down) {
        ChainGemm g = {ca.dgu, ca.inter, L.wdown, ca.dx, H, ca.dx, H, B, H, ca.inter, nullptr, 0.f, L.wdown8, L.sdown};
        chain_tile<T, EPI_RESID, false, 1, 6, WaitSharded, 4, false, W8>(g, blockIdx.x, ca.nwg_down, msm, WaitSharded{ctr, 0, ca.err, ca.naps, nullptr});
        publish_sc1(ctr, blockIdx.x);
        return;
    }
    const ChainLayer& Ln = ca.layers[ca.layer + 1];
    ChainGemm g = {ca.dx, H, Ln.wqkv, nullptr, 0, ca.dqkv, ca.qkv_ld, B, ca.qkv_n, H, Ln.attn_norm, ca.eps, Ln.wqkv8, Ln.sqkv};
    chain_tile<T, EPI_NONE, true, 4, 2, WaitSharded, 4, false, W8>(g, blockIdx.x - ca.nwg_down, (ca.qkv_n + 15) / 16, msm,
                                                                    WaitSharded{ctr, ca.nwg_down, ca.err, ca.naps, nullptr});
}

// ---- decode attention + o_proj(+residual) in ONE launch, 16-wave workgroups -------------------------------------------------
// Workgroups [0, heads*B): attention exactly as the stand-alone latency kernel (wave 0 = new token, 15 cache waves),
// output stored write-through. Workgroups [heads*B, +ntiles/2): two o_proj tiles each (8 waves per tile), whose WHOLE
// K slice (16 chunks per wave) goes in flight at entry and sits in registers while attention runs; then the fence-free
// hand-off (handoff.h) and ~2 us of work. heads*B + ntiles/2 <= 256 workgroups of <= 128 VGPRs: all resident, one per CU.
template <typename T, bool W8>
__global__ __launch_bounds__(CH_THREADS, 4) void attn_oproj16_k(DecAttnArgs at, ChainGemm g, int n_attn, int ntiles, int* counter, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    if ((int)blockIdx.x < n_attn) {
        const int b = blockIdx.x / at.d.heads, h = blockIdx.x - b * at.d.heads;
        decode_attention_body<T, CH_WAVES, true, NoWait, true, 0, true>(at, h, b, reinterpret_cast<float*>(msm));
        publish_sc1(counter, blockIdx.x);
    } else {
        chain_tile<T, EPI_RESID, false, 2, 2, WaitSharded, W8 ? 4 : 8, true, W8>(g, blockIdx.x - n_attn, ntiles, msm, WaitSharded{counter, n_attn, err, 1, nullptr});
    }
}

bool attn_oproj16_supported(const LlamaDims& d, int N, int K, int B) {
    const int ntiles = (N + 15) / 16;
    return B <= CH_MAXM && d.head_dim == 128 && K % 32 == 0 && N % 4 == 0 && (size_t)B * K <= 8192 &&
           d.heads * B + (ntiles + 1) / 2 <= 256;
}

void launch_attn_oproj16(int dtype, const DecAttnArgs& a, const GemmArgs& ga, int B, int* counter, int* err, hipStream_t s) {
    const int n_attn = a.d.heads * B, ntiles = (ga.N + 15) / 16;
    const bool w8 = ga.W8 && ga.wscale && ga.K % 64 == 0;
    ChainGemm g = {ga.X, ga.ldx, ga.W, ga.resid, ga.ldr, ga.out, ga.ldo, ga.M, ga.N, ga.K, nullptr, 0.f, ga.W8, ga.wscale};
    const size_t sm_gemm = (size_t)(CH_WAVES * 256 + CH_WAVES * CH_MAXM + 16) * 4 + (size_t)B * ga.K * 2;
    const size_t sm_att = decode_attention_smem_floats(CH_WAVES, a.d.max_len) * sizeof(float);
    const size_t smem = sm_gemm > sm_att ? sm_gemm : sm_att;
    dim3 grid(n_attn + (ntiles + 1) / 2), block(CH_THREADS);
    RDX_DISPATCH_T(dtype, T, {
        if (w8) hipLaunchKernelGGL((attn_oproj16_k<T, true>), grid, block, smem, s, a, g, n_attn, ntiles, counter, err);
        else hipLaunchKernelGGL((attn_oproj16_k<T, false>), grid, block, smem, s, a, g, n_attn, ntiles, counter, err);
    });
}

bool chain_supported(const LlamaDims& d, int inter, int B) {
    if (B > CH_MAXM || d.head_dim != 128 || d.hidden % 32 || inter % 32 || d.hidden % 16) return false;
    const size_t stage = (size_t)B * (inter > d.hidden ? inter : d.hidden) * 2;
    // staged activations: <= 2 x 1024 chunks of 4 elements for the K = hidden unit, <= 6 x 1024 for down_proj (K = inter)
    return stage <= 48 * 1024 && inter % 4 == 0 && (size_t)B * d.hidden <= 8192 && (size_t)B * inter <= 24576;
}

size_t chain_ctr_ints(int layers) { return (size_t)layers * HO_CTR_INTS; }

void launch_decode_chain(int dtype, ChainArgs ca, bool with_next_qkv, hipStream_t s) {
    ca.nwg_down = ca.hidden / 16;
    const int nwg_qkv = with_next_qkv ? ((ca.qkv_n + 15) / 16 + 3) / 4 : 0;
    const int kmax = ca.inter > ca.hidden ? ca.inter : ca.hidden;
    const size_t sm_gemm = (size_t)(CH_WAVES * 256 + CH_WAVES * CH_MAXM + 16) * 4 + (size_t)ca.B * kmax * 2;
    // ONE workgroup per CU: the register budget alone (62 VGPRs) would admit two, so reserve more than half of the LDS
    const size_t smem = sm_gemm > (size_t)84 * 1024 ? sm_gemm : (size_t)84 * 1024;
    dim3 grid(ca.nwg_down + nwg_qkv), block(CH_THREADS);
    RDX_DISPATCH_T(dtype, T, {
        static DevOnce attr_set;
        if (attr_set.first()) {
            hipFuncSetAttribute((const void*)decode_chain_k<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            hipFuncSetAttribute((const void*)decode_chain_k<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        }
        if (ca.w8) hipLaunchKernelGGL((decode_chain_k<T, true>), grid, block, smem, s, ca);
        else hipLaunchKernelGGL((decode_chain_k<T, false>), grid, block, smem, s, ca);
    });
}

}  // namespace rdx

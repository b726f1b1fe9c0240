// Chained decode-layer kernel (gfx950): every per-layer unit of the batch-1 decode step -- QKV GEMV, attention,
// o_proj, gate/up SwiGLU GEMV, down_proj -- for a RANGE of layers in ONE launch.
//
// Why: at batch 1 a decode step is 160 dependent launches of 8-30 us each, and every launch boundary leaves HBM idle for
// ~3-5 us (drain, dispatch ramp, first-load latency). Here the units are "roles" of one grid, ordered by blockIdx
// (layer-major, then qkv | attention | o_proj | gate/up | down), and the dependency between consecutive units is a
// counter hand-off (handoff.h) instead of a kernel boundary: a workgroup first puts two register batches of ITS weight
// tile in flight, then waits until the producer role has published the activations. While one unit drains, the
// workgroups of the next units are already resident with their weights streaming, so HBM stays busy.
//
// All roles are 1024-thread workgroups (<= 64 VGPRs: two per CU, 512 resident):
//   * qkv, gate/up (769 / 1376 output tiles): 4 tiles per workgroup, 4 waves split K of each tile (the shape of the
//     stand-alone 4-wave kernel); the RMS-normalised activation row is staged ONCE per workgroup in LDS for all 4 tiles;
//   * o_proj, down (256 tiles): one tile per workgroup, 16 waves split K;
//   * attention: attn_body.h, 16 waves (cached K rows go in flight before the wait).
// Arithmetic, rounding points and the fixed-order LDS reduction are those of skinny_body.h (same oracle parity).
// Liveness: a workgroup only waits on counters fed by lower-indexed workgroups; spins are bounded (handoff.h).
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "attn_body.h"
#include "handoff.h"
#include "skinny_body.h"   // swiglu()

namespace rdx {

constexpr int MG_WAVES = 16;
constexpr int MG_THREADS = MG_WAVES * 64;
constexpr int MG_MAXM = 2;                       // batch rows supported (LDS staging of [M][inter] activations)

enum { MG_QKV = 0, MG_ATT = 1, MG_O = 2, MG_GU = 3, MG_DOWN = 4, MG_NROLE = 5 };
constexpr int MG_CTR_STRIDE = HO_CTR_INTS;       // 8 shards x one 64-byte line per role counter

struct MegaGemm {                                // one weight-streaming unit
    const void* X; int ldx;
    const void* W;
    const void* resid; int ldr;
    void* out; int ldo;
    int M, N, K;
    const void* norm_w; float eps;
    const void* W8; const float* wscale;         // fp8 weights (64-deep fragment order) + per-row scale, see skinny_body.h
};

template <typename T, int EPI, bool NORM, int SUB, int XL, typename WaitFn, int U = 4, bool RESID_EARLY = false, bool W8 = false>
__device__ __forceinline__ void mega_tile(const MegaGemm& a, const int wg, const int ntiles, unsigned char* smem, WaitFn wait_inputs) {
    typedef typename Vec8<T>::type V8;
    constexpr int WPS = MG_WAVES / SUB;           // waves per tile
    // U chunks per register batch, two batches in flight: U = 4 keeps every role of the chained kernel <= 64 VGPRs
    float* red = reinterpret_cast<float*>(smem);                        // [MG_WAVES][256]
    float* ssq = red + MG_WAVES * 256;                                  // [MG_WAVES][MG_MAXM]
    float* rstd_s = ssq + MG_WAVES * MG_MAXM;                           // [MG_MAXM] (+pad)
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
    // XL = chunks of 4 elements per thread: M*K/4 <= XL*1024 (checked by mega_supported)
    const int K4 = K >> 2, total4 = a.M * K4;
    unsigned long long xr[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int c = threadIdx.x + i * MG_THREADS;
        xr[i] = 0ull;
        if (c < total4) { const int m = c / K4, k4 = c - m * K4; xr[i] = ld8_agent(X + (size_t)m * a.ldx + (size_t)k4 * 4); }
    }
    if (NORM) {
        float ss[MG_MAXM];
#pragma unroll
        for (int m = 0; m < MG_MAXM; ++m) ss[m] = 0.f;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int c = threadIdx.x + i * MG_THREADS;
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float f = tof<T>(from_bits16<T>((unsigned short)(xr[i] >> (16 * j)))); t += f * f; }
            const int m = c < total4 ? c / K4 : 0;                       // padding chunks are zero
#pragma unroll
            for (int mm = 0; mm < MG_MAXM; ++mm) ss[mm] += (m == mm) ? t : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MG_MAXM; ++m) {
            const float t = wave_sum(ss[m]);
            if (lane == 0) ssq[wa * MG_MAXM + m] = t;
        }
        __syncthreads();
        if (threadIdx.x < MG_MAXM) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < MG_WAVES; ++i) t += ssq[i * MG_MAXM + threadIdx.x];
            rstd_s[threadIdx.x] = rsqrtf(t / (float)K + a.eps);
        }
        __syncthreads();
    }
    {
        const T* NW = reinterpret_cast<const T*>(a.norm_w);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int c = threadIdx.x + i * MG_THREADS;
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

// OCC = waves per SIMD the register budget is sized for: 8 -> <= 64 VGPRs, two workgroups per CU (attention keeps a
// small K window and loads V late); 4 -> <= 128 VGPRs, one workgroup per CU (attention as in the stand-alone kernel)
template <typename T, int OCC, bool WITH_ATT, bool W8 = false>
__global__ __launch_bounds__(MG_THREADS, OCC) void decode_layers_k(MegaArgs ma) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    const int per_layer = ma.nwg[0] + ma.nwg[1] + ma.nwg[2] + ma.nwg[3] + ma.nwg[4];
    const int bid = blockIdx.x + ma.blk_offset;             // the launch may start in the middle of its first layer
    const int li = bid / per_layer;                         // layer index inside this launch
    int rb = bid - li * per_layer;
    const int l = ma.layer0 + li;
    const MegaLayer& L = ma.layers[l];
    int* ctr = ma.ctr + (size_t)l * MG_NROLE * MG_CTR_STRIDE;
    int* prev_down = ctr - MG_NROLE * MG_CTR_STRIDE + MG_DOWN * MG_CTR_STRIDE;
    const int H = ma.d.hidden, B = ma.B;
    long long* tr = ma.trace ? ma.trace + (size_t)blockIdx.x * 4 : nullptr;
    if (tr && threadIdx.x == 0) tr[0] = (long long)__builtin_amdgcn_s_memrealtime();
#define MG_DONE(role) do { if (tr && threadIdx.x == 0) { tr[2] = (long long)__builtin_amdgcn_s_memrealtime(); tr[3] = (role); } } while (0)

    if (rb < ma.nwg[MG_QKV]) {
        MegaGemm g = {ma.dx, H, L.wqkv, nullptr, 0, ma.dqkv, ma.d.qkv_ld, B, ma.qkv_n, H, L.attn_norm, ma.eps, L.wqkv8, L.sqkv};
        // first layer of the launch: the kernel boundary already ordered it after the previous launch
        mega_tile<T, EPI_NONE, true, 4, 2, WaitSharded, 4, false, W8>(g, rb, ma.tiles[MG_QKV], msm,
                                           WaitSharded{prev_down, li ? ma.nwg[MG_DOWN] : 0, ma.err, ma.naps, tr});
        publish_sc1(ctr + MG_QKV * MG_CTR_STRIDE, rb);
        MG_DONE(MG_QKV);
        return;
    }
    rb -= ma.nwg[MG_QKV];
    if (WITH_ATT && rb < ma.nwg[MG_ATT]) {
        DecAttnArgs at;
        at.d = ma.d; at.qkv = ma.dqkv; at.lbq = L.lbq; at.lbv = L.lbv; at.cos_t = ma.cos_t; at.sin_t = ma.sin_t; at.cur_rope = ma.cur_rope;
        at.pos = ma.pos; at.slot_b = ma.slot_b; at.key_mask = ma.key_mask; at.kcache = L.kcache; at.vcache = L.vcache; at.out = ma.datt;
        at.trace = ma.trace ? ma.trace + (size_t)gridDim.x * 4 + (size_t)l * 8 : nullptr;
        const int b = rb / ma.d.heads, h = rb - b * ma.d.heads;
        const WaitSharded wq{ctr + MG_QKV * MG_CTR_STRIDE, (li || ma.r_begin != MG_ATT) ? ma.nwg[MG_QKV] : 0, ma.err, ma.naps, tr};
        if (OCC == 8) decode_attention_body<T, MG_WAVES, true, WaitSharded, false, 1, true>(at, h, b, reinterpret_cast<float*>(msm), wq);
        else decode_attention_body<T, MG_WAVES, true, WaitSharded, true, 0, true>(at, h, b, reinterpret_cast<float*>(msm), wq);
        publish_sc1(ctr + MG_ATT * MG_CTR_STRIDE, rb);
        MG_DONE(MG_ATT);
        return;
    }
    rb -= ma.nwg[MG_ATT];
    if (rb < ma.nwg[MG_O]) {
        MegaGemm g = {ma.datt, H, L.wo, ma.dx, H, ma.dx, H, B, H, H, nullptr, 0.f, L.wo8, L.so};
        mega_tile<T, EPI_RESID, false, 1, 2, WaitSharded, 4, false, W8>(g, rb, ma.tiles[MG_O], msm, WaitSharded{ctr + MG_ATT * MG_CTR_STRIDE, (li || ma.r_begin != MG_O) ? ma.nwg[MG_ATT] : 0, ma.err, ma.naps, tr});
        publish_sc1(ctr + MG_O * MG_CTR_STRIDE, rb);
        MG_DONE(MG_O);
        return;
    }
    rb -= ma.nwg[MG_O];
    if (rb < ma.nwg[MG_GU]) {
        MegaGemm g = {ma.dx, H, L.wgu, nullptr, 0, ma.dgu, ma.inter, B, 2 * ma.inter, H, L.mlp_norm, ma.eps, L.wgu8, L.sgu};
        mega_tile<T, EPI_SILU_MUL, true, 4, 2, WaitSharded, 4, false, W8>(g, rb, ma.tiles[MG_GU], msm, WaitSharded{ctr + MG_O * MG_CTR_STRIDE, (li || ma.r_begin != MG_GU) ? ma.nwg[MG_O] : 0, ma.err, ma.naps, tr});
        publish_sc1(ctr + MG_GU * MG_CTR_STRIDE, rb);
        MG_DONE(MG_GU);
        return;
    }
    rb -= ma.nwg[MG_GU];
    {
        MegaGemm g = {ma.dgu, ma.inter, L.wdown, ma.dx, H, ma.dx, H, B, H, ma.inter, nullptr, 0.f, L.wdown8, L.sdown};
        mega_tile<T, EPI_RESID, false, 1, 6, WaitSharded, 4, false, W8>(g, rb, ma.tiles[MG_DOWN], msm, WaitSharded{ctr + MG_GU * MG_CTR_STRIDE, (li || ma.r_begin != MG_DOWN) ? ma.nwg[MG_GU] : 0, ma.err, ma.naps, tr});
        publish_sc1(ctr + MG_DOWN * MG_CTR_STRIDE, rb);
        MG_DONE(MG_DOWN);
    }
}

// ---- decode attention + o_proj(+residual) in ONE launch, 16-wave workgroups -------------------------------------------------
// Workgroups [0, heads*B): attention exactly as the stand-alone latency kernel (wave 0 = new token, 15 cache waves),
// output stored write-through. Workgroups [heads*B, +ntiles/2): two o_proj tiles each (8 waves per tile), whose WHOLE
// K slice (16 chunks per wave) goes in flight at entry and sits in registers while attention runs; then the fence-free
// hand-off (handoff.h) and ~2 us of work. heads*B + ntiles/2 <= 256 workgroups of <= 128 VGPRs: all resident, one per CU.
template <typename T, bool W8>
__global__ __launch_bounds__(MG_THREADS, 4) void attn_oproj16_k(DecAttnArgs at, MegaGemm g, int n_attn, int ntiles, int* counter, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    if ((int)blockIdx.x < n_attn) {
        const int b = blockIdx.x / at.d.heads, h = blockIdx.x - b * at.d.heads;
        decode_attention_body<T, MG_WAVES, true, NoWait, true, 0, true>(at, h, b, reinterpret_cast<float*>(msm));
        publish_sc1(counter, blockIdx.x);
    } else {
        mega_tile<T, EPI_RESID, false, 2, 2, WaitSharded, W8 ? 4 : 8, true, W8>(g, blockIdx.x - n_attn, ntiles, msm, WaitSharded{counter, n_attn, err, 1, nullptr});
    }
}

bool attn_oproj16_supported(const LlamaDims& d, int N, int K, int B) {
    const int ntiles = (N + 15) / 16;
    return B <= MG_MAXM && d.head_dim == 128 && K % 32 == 0 && N % 4 == 0 && (size_t)B * K <= 8192 &&
           d.heads * B + (ntiles + 1) / 2 <= 256;
}

void launch_attn_oproj16(int dtype, const DecAttnArgs& a, const GemmArgs& ga, int B, int* counter, int* err, hipStream_t s) {
    const int n_attn = a.d.heads * B, ntiles = (ga.N + 15) / 16;
    const bool w8 = ga.W8 && ga.wscale && ga.K % 64 == 0;
    MegaGemm g = {ga.X, ga.ldx, ga.W, ga.resid, ga.ldr, ga.out, ga.ldo, ga.M, ga.N, ga.K, nullptr, 0.f, ga.W8, ga.wscale};
    const size_t sm_gemm = (size_t)(MG_WAVES * 256 + MG_WAVES * MG_MAXM + 16) * 4 + (size_t)B * ga.K * 2;
    const size_t sm_att = decode_attention_smem_floats(MG_WAVES, a.d.max_len) * sizeof(float);
    const size_t smem = sm_gemm > sm_att ? sm_gemm : sm_att;
    dim3 grid(n_attn + (ntiles + 1) / 2), block(MG_THREADS);
    RDX_DISPATCH_T(dtype, T, {
        if (w8) hipLaunchKernelGGL((attn_oproj16_k<T, true>), grid, block, smem, s, a, g, n_attn, ntiles, counter, err);
        else hipLaunchKernelGGL((attn_oproj16_k<T, false>), grid, block, smem, s, a, g, n_attn, ntiles, counter, err);
    });
}

bool mega_supported(const LlamaDims& d, int inter, int B) {
    if (B > MG_MAXM || d.head_dim != 128 || d.hidden % 32 || inter % 32 || d.hidden % 16) return false;
    const size_t stage = (size_t)B * (inter > d.hidden ? inter : d.hidden) * 2;
    // staged activations: <= 2 x 1024 chunks of 4 elements for the K = hidden units, <= 6 x 1024 for down_proj (K = inter)
    return stage <= 48 * 1024 && inter % 4 == 0 && (size_t)B * d.hidden <= 8192 && (size_t)B * inter <= 24576;
}

size_t mega_ctr_ints(int layers) { return (size_t)layers * MG_NROLE * MG_CTR_STRIDE; }

static void mega_fill(MegaArgs& ma) {
    ma.tiles[MG_QKV] = (ma.qkv_n + 15) / 16;  ma.nwg[MG_QKV] = (ma.tiles[MG_QKV] + 3) / 4;
    ma.tiles[MG_ATT] = ma.d.heads * ma.B;     ma.nwg[MG_ATT] = ma.tiles[MG_ATT];
    ma.tiles[MG_O] = ma.d.hidden / 16;        ma.nwg[MG_O] = ma.tiles[MG_O];
    ma.tiles[MG_GU] = (2 * ma.inter) / 16;    ma.nwg[MG_GU] = (ma.tiles[MG_GU] + 3) / 4;
    ma.tiles[MG_DOWN] = ma.d.hidden / 16;     ma.nwg[MG_DOWN] = ma.tiles[MG_DOWN];
}

void launch_decode_roles(int dtype, MegaArgs ma, int R0, int R1, int occ, hipStream_t s) {
    mega_fill(ma);
    const int l0 = R0 / MG_NROLE, r0 = R0 % MG_NROLE, l1 = (R1 - 1) / MG_NROLE, r1 = (R1 - 1) % MG_NROLE;
    bool with_att = false;
    for (int R = R0; R < R1; ++R) with_att |= (R % MG_NROLE) == MG_ATT;
    const int per_layer = ma.nwg[0] + ma.nwg[1] + ma.nwg[2] + ma.nwg[3] + ma.nwg[4];
    int off = 0, tail = 0;
    for (int r = 0; r < r0; ++r) off += ma.nwg[r];
    for (int r = r1 + 1; r < MG_NROLE; ++r) tail += ma.nwg[r];
    ma.layer0 = l0; ma.r_begin = r0; ma.blk_offset = off;
    const int kmax = ma.inter > ma.d.hidden ? ma.inter : ma.d.hidden;
    const size_t sm_gemm = (size_t)(MG_WAVES * 256 + MG_WAVES * MG_MAXM + 16) * 4 + (size_t)ma.B * kmax * 2;
    const size_t sm_att = with_att ? decode_attention_smem_floats(MG_WAVES, ma.d.max_len) * sizeof(float) : 0;
    const size_t smem = sm_gemm > sm_att ? sm_gemm : sm_att;
    dim3 grid((l1 - l0 + 1) * per_layer - off - tail), block(MG_THREADS);
    RDX_DISPATCH_T(dtype, T, {
        if (!with_att && occ == 8) hipLaunchKernelGGL((decode_layers_k<T, 8, false>), grid, block, smem, s, ma);
        else if (!with_att) {
            // ONE workgroup per CU: two resident 1024-thread workgroups per CU measured ~25 % slower in every unit, and the
            // register budget alone (62 VGPRs) would admit two, so reserve more than half of the LDS
            const size_t big = smem > (size_t)84 * 1024 ? smem : (size_t)84 * 1024;
            static bool attr_set[2] = {false, false};
            if (!attr_set[dtype & 1]) {
                hipFuncSetAttribute((const void*)decode_layers_k<T, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
                hipFuncSetAttribute((const void*)decode_layers_k<T, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
                attr_set[dtype & 1] = true;
            }
            if (ma.w8) hipLaunchKernelGGL((decode_layers_k<T, 4, false, true>), grid, block, big, s, ma);
            else hipLaunchKernelGGL((decode_layers_k<T, 4, false>), grid, block, big, s, ma);
        }
        else if (occ == 8) hipLaunchKernelGGL((decode_layers_k<T, 8, true>), grid, block, smem, s, ma);
        else hipLaunchKernelGGL((decode_layers_k<T, 4, true>), grid, block, smem, s, ma);
    });
}

void launch_decode_layers(int dtype, MegaArgs ma, int nlayers, int occ, hipStream_t s) {
    launch_decode_roles(dtype, ma, ma.layer0 * MG_NROLE, (ma.layer0 + nlayers) * MG_NROLE, occ, s);
}


}  // namespace rdx

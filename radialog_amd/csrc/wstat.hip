// Weight-stationary GEMM for ONE prompt's prefill (16 < M <= a few hundred rows; QKV, o_proj, gate/up, down of the Llama layer), gfx950.
//
// Why: at M = 160 the projections are weight-stream bound (404 MB per layer, 64 us at 6.3 TB/s; their MFMA time is 26 us), but the
// 128 x 128 LDS-DMA tiles pad M to 256, read every weight panel twice and move 32 KiB through the CU's 64 B/clk vector-memory path per
// 0.7-us step: gate/up ran at 2.1 TB/s (85 us; the vendor GEMM: 54 us, tools/blas_yardstick.py). Per CU that path, not HBM and not the
// matrix pipe, is the limit: whatever is not stationary in the CU has to come through it.
//
// Here the WEIGHTS are stationary and read from HBM exactly once: a workgroup owns NT output tiles of 16 columns; wave w keeps their
// MFMA A fragments for its K range (KC / WAVES chunks of 32) in registers (NT x CPW x 4 VGPRs: half the CU's register file at
// NT = 2, K = 4096, 8 waves). The ACTIVATIONS stream past them, row tile by row tile, out of L2: they arrive fragment-packed
// ([k / 32][row tile][lane][8] -- written so by rmsnorm_k<T, 3>, the attention kernel and this kernel's SwiGLU epilogue), so a wave's
// B fragment is one contiguous KiB, fetched through a register ring that is refilled one load per consumed fragment (issue order =
// consume order, the waits stay counted). Per row tile a wave does NT x CPW MFMAs and drops its 16 x 16 fp32 partials in one of four
// LDS buffers; there is NO workgroup barrier in the loop (LDS counters instead, see below): every wave adds the WAVES partials of the
// PREVIOUS row tile for its own 64 outputs in a fixed order (deterministic, no atomics on data) and runs the epilogue while its next
// fragments are landing. The same rounding points as the other GEMM kernels (skinny_body.h / gemm.hip epilogues).
//
// Cost model: a workgroup loads NT x K x 32 B of weights (HBM), then streams M x K x 2 B of activations through the
// CU's L1 (64 B/clk: 8.9 us at M = 160, K = 4096) whatever NT is -- so NT is as large as the registers allow.
#include <type_traits>
#include <algorithm>
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"   // swiglu()

namespace rdx {

template <typename T, int EPI, int NT, int WAVES, int CPW, int RING>
__global__ __launch_bounds__(WAVES * 64) void wstat_k(GemmArgs a) {
    static_assert(RING <= CPW, "the ring refills at most one row tile ahead");
    extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
    float* red = reinterpret_cast<float*>(smw);                     // [NBUF][WAVES][NT][256]
    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = (a.N + 15) >> 4, MT = a.mtiles;
    const int t0 = (int)blockIdx.x * NT;
    const int KC = a.K >> 5;
    const int c0 = wa * CPW;                                        // KC == WAVES * CPW (wstat_supported): every wave owns exactly CPW chunks
    long long* trc = (a.trace && threadIdx.x == 0) ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
#define WS_T(i) do { if (trc) trc[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    WS_T(0);

    // ---- stationary weights: the cold HBM loads go first
    u4 wreg[NT][CPW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const u4* wp = reinterpret_cast<const u4*>(a.W) + ((size_t)min(t0 + nt, ntiles - 1) * KC + c0) * 64 + lane;
#pragma unroll
        for (int j = 0; j < CPW; ++j) wreg[nt][j] = ldg16_nt(wp + j * 64);
    }
    // ---- activation ring: fragment (chunk c0 + j, row tile mt) is one contiguous KiB; the loads walk (mt, j) in consume order, so ONE
    // running pointer serves them all: + one chunk (MT KiB), or back to chunk 0 of the next row tile
    const u4* lp = reinterpret_cast<const u4*>(a.X) + (size_t)c0 * MT * 64 + lane;          // (chunk c0, row tile 0)
    const long cstride = (long)MT * 64, wrap = 64 - (long)(CPW - 1) * MT * 64;              // in 16-byte units
    u4 ring[RING];
#pragma unroll
    for (int j = 0; j < RING; ++j) {
        ring[j] = ldg16(lp);
        lp += (j == CPW - 1) ? wrap : cstride;
        __builtin_amdgcn_sched_barrier(0);
    }

    T* out = reinterpret_cast<T*>(a.out);
    // ---- no workgroup barrier in the loop. A barrier per row tile made every wave wait for the slowest one and then for the epilogue, and
    // while waves wait nobody consumes (and therefore re-issues) activation loads: the CU's L1 idled for half of every row tile (in-kernel
    // timeline, tools/wstat_trace.py: K loop 0.45 us, barrier wait 0.6 us, epilogue 0.2-0.6 us per 0.87 us of L1 transfer).
    // Instead the partials go to one of NBUF LDS buffers and the waves only count: posted[b] += 1 after a wave's partial of row tile mt is in
    // buffer b = mt % NBUF; every wave then runs the epilogue of the PREVIOUS row tile for its own 64 of the 256 NT outputs (its residual was
    // requested a whole K loop earlier), waiting -- almost never -- until all WAVES partials are posted, and counts released[b] += 1 when it
    // has read them; a buffer is rewritten NBUF row tiles later, once all waves have released it. Waves drift apart by up to NBUF - 1 row tiles.
    constexpr int NBUF = 4;
    __shared__ int ctr[2 * NBUF];                                   // posted[NBUF], released[NBUF]: cumulative counts (static LDS: ds_* instructions, no FLAT)
    if (threadIdx.x < 2 * NBUF) ctr[threadIdx.x] = 0;
    __syncthreads();
    auto wait_ge = [&](int idx, int target) {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctr[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");                              // LDS executes a wave's operations in order: what follows sees what was counted
    };
    // epilogue slice of this wave: output o = 64 wave + lane of the NT x 256 block -> (tile e_nt, row e_ml, column e_nl)
    const int e_o = wa * 64 + lane;
    const bool e_on = e_o < NT * 256;                               // NT = 1: waves 0-3
    const int e_nt = e_o >> 8, e_idx = e_o & 255, e_ml = e_idx >> 4, e_nl = e_idx & 15;
    const int e_t = t0 + e_nt, e_n = e_t * 16 + e_nl;
    const bool e_col = e_on && e_t < ntiles && e_n < a.N;
    const float e_bias = (a.bias && e_on) ? a.bias[min(e_n, a.N - 1)] : 0.f;

    // ring slot of chunk j = j % RING in EVERY row tile: a row tile is VP = RING-multiple virtual steps, the VP - CPW steps at its end are
    // bubbles (no fragment, no MFMA) that only refill their slot for the next row tile (K = 11008: 43 chunks on a ring of 8 -> 48 steps)
    constexpr int VP = (CPW + RING - 1) / RING * RING;
    auto kloop = [&](v4f (&acc)[NT], auto pf_tag) {
        constexpr bool PF = decltype(pf_tag)::value;                 // false: the kernel's last row tile, nothing left to fetch
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            if (j < CPW) {
                const u4 xv = ring[j % RING];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(as_vec8<T>(wreg[nt][j]), as_vec8<T>(xv), acc[nt]);
            }
            const int q = j + RING < CPW ? j + RING : j + RING - VP;                       // chunk to fetch: of this row tile, or of the next
            if (j + RING < CPW || (PF && j + RING >= VP)) {
                ring[j % RING] = ldg16(lp);
                lp += (q == CPW - 1) ? wrap : cstride;
            }
            __builtin_amdgcn_sched_barrier(0);               // consume-j / refill-j order: the waits stay counted
        }
    };
    auto load_res = [&](int mt) -> T {                              // raw bits; converted in the epilogue so that the wait in between stays counted
        const int e_m = mt * 16 + e_ml;
        if (EPI == EPI_RESID && e_on) return reinterpret_cast<const T*>(a.resid)[(size_t)min(e_m, a.M - 1) * a.ldr + min(e_n, a.N - 1)];
        return fromf<T>(0.f);
    };
    auto post = [&](int mt, const v4f (&acc)[NT]) {
        const int b_ = mt % NBUF;
        if (mt >= NBUF) wait_ge(NBUF + b_, WAVES * (mt / NBUF));
        // D[n_local = 4 g + reg][m_local = r] -> red[buf][wave][nt][m_local * 16 + n_local]
        float* rb = red + (size_t)b_ * (WAVES * NT * 256);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<float4*>(&rb[(wa * NT + nt) * 256 + r * 16 + g * 4]) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the partial is in LDS before the count moves
        if (lane == 0) __hip_atomic_fetch_add(&ctr[b_], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto epilogue = [&](int mt, T res) {
        const int b_ = mt % NBUF;
        wait_ge(b_, WAVES * (mt / NBUF + 1));
        const float* rb = red + (size_t)b_ * (WAVES * NT * 256);
        float v = 0.f;
        if (e_on) {
#pragma unroll
            for (int i = 0; i < WAVES; ++i) v += rb[(i * NT + e_nt) * 256 + e_idx];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // the partials have been read
        if (lane == 0) __hip_atomic_fetch_add(&ctr[NBUF + b_], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v += e_bias;
        const int e_m = mt * 16 + e_ml;
        const bool ok = e_col && e_m < a.M;
        if (EPI == EPI_NONE) {
            if (ok) out[(size_t)e_m * a.ldo + e_n] = fromf<T>(v);
        } else if (EPI == EPI_RESID) {
            if (ok) out[(size_t)e_m * a.ldo + e_n] = fromf<T>(tof<T>(res) + rnd<T>(v));
        } else if (EPI == EPI_SILU_MUL) {
            // columns 0-7 of a tile are gate rows, 8-15 the matching up rows: the partner sits 8 lanes away in the same DPP row
            const float u = dpp_mov<DPP_ROR8>(v);
            if (a.out_packed) {
                // fragment-packed for the down projection: this tile's 8 outputs k = 8 t .. + 8 of row m are one lane's 16-byte piece
                // of fragment (k / 32, row tile); rows >= M are zero-filled
                if (e_on && e_nl < 8 && e_t < ntiles)
                    out[((((size_t)(e_t >> 2) * MT + mt) * 64 + (e_t & 3) * 16 + e_ml) << 3) + e_nl] =
                        e_m < a.M ? fromf<T>(swiglu<T>(v, u)) : fromf<T>(0.f);
            } else if (e_nl < 8 && ok) out[(size_t)e_m * a.ldo + e_t * 8 + e_nl] = fromf<T>(swiglu<T>(v, u));
        }
    };

    v4f acc[NT];
    T res_prev = fromf<T>(0.f);
    for (int mt = 0; mt + 1 < MT; ++mt) {
        const T res_cur = load_res(mt);                             // needed one row tile from now
        if (mt == 4) WS_T(4);
        kloop(acc, std::true_type{});
        if (mt == 0) WS_T(1);
        if (mt == 4) WS_T(5);
        post(mt, acc);
        if (mt == 4) WS_T(6);
        if (mt > 0) epilogue(mt - 1, res_prev);
        if (mt == 4) WS_T(7);
        res_prev = res_cur;
    }
    WS_T(2);
    {
        const T res_cur = load_res(MT - 1);
        kloop(acc, std::false_type{});
        post(MT - 1, acc);
        if (MT > 1) epilogue(MT - 2, res_prev);
        epilogue(MT - 1, res_cur);
    }
    WS_T(3);
#undef WS_T
}

// shapes: activations fragment-packed in `mtiles` row tiles (xpacked 3), K % 32 == 0; K = 4096 -> 16 chunks per wave, K = 11008 -> 43 (the Vicuna-7B projections; other K keep the tile GEMMs)
static int wstat_cfg(const GemmArgs& a, int epi) {
    if (a.xpacked != 3 || a.mtiles <= 0 || a.M > a.mtiles * 16 || (a.K & 31) || a.norm_w) return 0;
    if (epi != EPI_NONE && epi != EPI_RESID && epi != EPI_SILU_MUL) return 0;
    if (epi != EPI_SILU_MUL && a.out_packed) return 0;
    const int KC = a.K >> 5;
    if (KC == 8 * 16) return 1;                 // K = 4096: 8 waves x 16 chunks
    if (KC == 8 * 43) return 2;                 // K = 11008: 8 waves x 43 chunks
    return 0;
}

bool wstat_supported(const GemmArgs& a, int epi) {
    return wstat_cfg(a, epi) != 0;
}

template <typename T, int EPI, int NT, int WAVES, int CPW, int RING>
static void launch_ws1(const GemmArgs& a, hipStream_t s) {
    const int ntiles = (a.N + 15) >> 4;
    const size_t smem = (size_t)4 * WAVES * NT * 256 * sizeof(float);
    static DevOnce attr;
    if (smem > 48 * 1024 && attr.first()) { (void)hipFuncSetAttribute((const void*)wstat_k<T, EPI, NT, WAVES, CPW, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); }
    hipLaunchKernelGGL((wstat_k<T, EPI, NT, WAVES, CPW, RING>), dim3((ntiles + NT - 1) / NT), dim3(WAVES * 64), smem, s, a);
}

template <typename T, int EPI>
static void launch_ws_T(const GemmArgs& a, int cfg, hipStream_t s) {
    const int ntiles = (a.N + 15) >> 4;
    if (cfg == 2) { launch_ws1<T, EPI, 1, 8, 43, 8>(a, s); return; }
    // one tile per workgroup while that still gives every CU one (o_proj: 256 tiles); two otherwise
    const int nt = ntiles <= 256 ? 1 : 2;
    if (nt == 1) launch_ws1<T, EPI, 1, 8, 16, 16>(a, s);
    else launch_ws1<T, EPI, 2, 8, 16, 16>(a, s);
}

void launch_wstat(int dtype, const GemmArgs& a, int epi, hipStream_t s) {
    const int cfg = wstat_cfg(a, epi);
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_NONE) launch_ws_T<T, EPI_NONE>(a, cfg, s);
        else if (epi == EPI_RESID) launch_ws_T<T, EPI_RESID>(a, cfg, s);
        else launch_ws_T<T, EPI_SILU_MUL>(a, cfg, s);
    });
}

}  // namespace rdx

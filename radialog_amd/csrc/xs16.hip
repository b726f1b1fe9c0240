// Batch 3-16 decode projections (gfx950): ONE MFMA row tile of 16 rows -- five launches per layer instead of the seven of the 32-row
// family (xstat32.hip). The reference's own evaluation batch is 12 (test.py:279,:344); its decoder layer is
// modeling_llama_imgemb.py:266-318 (RMSNorm -> q/k/v -> attention -> o_proj + residual -> RMSNorm -> SwiGLU MLP + residual).
//
// What the 32-row family pays for at 3-16 rows: (1) half of every activation fragment, MFMA and start-up fetch is padding; (2) the two
// 256-tile projections (o_proj, down_proj) run K-split over fp32 slabs because [32][11008] activations do not fit a workgroup, and the
// slab combine + residual + RMSNorm is a 5-us launch of its own in front of QKV and of gate/up. At 16 rows neither is needed:
//
//   xstat16_k   QKV / gate-up / lm_head, K = 4096: activation-stationary like xstat32_k (256 persistent 8-wave workgroups, wave w owns
//               k in [512 w, 512 w + 512), 16 fragments = 64 VGPRs), two output tiles per trip (a 32-fragment weight ring: 256 KiB per CU in
//               flight), and the RMSNorm is its PROLOGUE: the workgroup reads the 16 residual-stream rows row-major (a wave's row segment is
//               one contiguous KiB), takes the row statistics (lanes -> waves through LDS, fixed order), scales and rounds exactly like
//               rmsnorm4096_k -- T(w * T(x * rstd)) -- and turns the rows into MFMA B fragments through a wave-private, conflict-free LDS
//               patch (ds_write_b128 / ds_read_b128; the direct gather from row-major memory is 16 rows x 64 B per load instruction). The
//               rows are requested first and the first weight ring right behind them, so the arithmetic runs under the weights' flight.
//   xrow16_k    o_proj / down_proj (+ residual): one 16-wave workgroup per output tile, K split over its waves, weight fragments (HBM,
//               non-temporal) AND activation fragments (fragment-packed by the producer's epilogue, L2) streamed through one register ring,
//               fixed-order LDS reduction, residual epilogue writing the final rows -- no slabs, nothing left for a later launch.
// Same rounding points as the GEMV / 32-row families (skinny_body.h): only the fp32 accumulation order differs.
#include <type_traits>
#include <algorithm>
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"   // swiglu()

namespace rdx {

constexpr int X16_WAVES = 8, X16_THREADS = 512, X16_K = 4096, X16_CPW = X16_K / 32 / X16_WAVES;   // 16 chunks of 32 per wave
constexpr int X16_TPI = 2, X16_RING = X16_TPI * X16_CPW;                                             // 32 fragments = 32 KiB per wave in flight
constexpr int X16_TR_ROW = 1024 + 16;       // bytes per row of a wave's transpose patch: 16 rows land 4 banks apart (b128 accesses, 16 lanes per pass)
constexpr int X16_TR_WAVE = 16 * X16_TR_ROW;
constexpr size_t X16_SMEM = (size_t)X16_WAVES * X16_TR_WAVE + 1024;                                   // patches (the partial buffers alias them) + statistics

template <typename T, int EPI>
__global__ __launch_bounds__(X16_THREADS) void xstat16_k(GemmArgs a) {
    typedef typename Vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
    float* red = reinterpret_cast<float*>(smx);                                   // aliases the transpose patches: a barrier separates the two uses
    float* ssq = reinterpret_cast<float*>(smx + (size_t)X16_WAVES * X16_TR_WAVE);  // [8 waves][16 rows]
    float* rstd_s = ssq + X16_WAVES * 16;                                          // [16]

    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = (a.N + 15) >> 4;
    const int G = gridDim.x;
    const int ngroups = (ntiles + X16_TPI - 1) / X16_TPI;
    const int nit = (ngroups - (int)blockIdx.x + G - 1) / G;
    if (nit <= 0) return;

    const u4* wbase = reinterpret_cast<const u4*>(a.W) + (size_t)(wa * X16_CPW) * 64;   // wave-uniform
    auto tile_ptr = [&](int t) { return wbase + (size_t)min(t, ntiles - 1) * (X16_K / 32) * 64; };

    u4 ring[X16_RING];
    auto first_ring = [&](int q) {
        const u4* wp = tile_ptr((int)blockIdx.x * X16_TPI + q);
#pragma unroll
        for (int j = 0; j < X16_CPW; ++j) {
            ring[q * X16_CPW + j] = ldg16_nt(wp + (unsigned)(j * 64 + lane));
            __builtin_amdgcn_sched_barrier(0);               // issue order = consume order (the loop's counted waits rely on it)
        }
    };
    // ---- activations -> B fragments xf[c]: column = row r of X, k = 512 wa + 32 c + 8 g .. + 8 ------------------------------------------
    const T* X = reinterpret_cast<const T*>(a.X);
    u4 xf[X16_CPW];
    u4 nw4;
    {
        // row-major rows, this wave's k range: lane holds k = 512 wa + 8 lane .. + 8 of every row. These 17 KiB per wave are
        // requested FIRST (whole-KiB row segments, L2 hits after the first workgroup of an XCD) and the weight ring right behind them, so the
        // prologue's arithmetic runs under the flight of the first weights and its wait (vmcnt = ring) does not drain them
        const T* xw = X + wa * 512 + lane * 8;
        nw4 = ldg16(reinterpret_cast<const T*>(a.norm_w) + wa * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < 16; ++i) xf[i] = ldg16(xw + (size_t)min(i, a.M - 1) * a.ldx);      // unconditional (a select would serialise the loads behind
                                                                                                // vmcnt(0)); rows >= M repeat the last row: never stored
        __builtin_amdgcn_sched_barrier(0);
    }
    // the first tile's weights; the second tile's follow BEHIND the prologue (its scaling temporaries and a 32-fragment ring do not fit the register
    // file together: 56 bytes of scratch when both were issued up front)
    first_ring(0);
    {
        // LlamaRMSNorm (:85-93): fp32 mean of squares per row -- lanes (DPP), then waves through LDS in wave order
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const V8 xv = as_vec8<T>(xf[i]);
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = tof<T>(xv[j]); ss += f * f; }
            ss = wave_sum(ss);
            if (lane == 0) ssq[wa * 16 + i] = ss;
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < X16_WAVES; ++i) t += ssq[i * 16 + threadIdx.x];
            rstd_s[threadIdx.x] = rsqrtf(t / (float)X16_K + a.eps);
        }
        __syncthreads();
        unsigned char* patch = smx + (size_t)wa * X16_TR_WAVE;
        const V8 nw = as_vec8<T>(nw4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float rs = rstd_s[i];
            // opaque to the optimiser: it would otherwise keep the 128 fp32 conversions of the statistics pass alive for this one
            asm volatile("" : "+v"(xf[i].x), "+v"(xf[i].y), "+v"(xf[i].z), "+v"(xf[i].w));
            V8 xv = as_vec8<T>(xf[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = fromf<T>(tof<T>(nw[j]) * rnd<T>(tof<T>(xv[j]) * rs));     // weight * (x * rsqrt(var + eps)).to(dtype)
            *reinterpret_cast<u4*>(patch + i * X16_TR_ROW + lane * 16) = as_u4<T>(xv);
        }
        __builtin_amdgcn_wave_barrier();        // the patch is wave-private: LDS serves a wave's operations in order
#pragma unroll
        for (int c = 0; c < X16_CPW; ++c) xf[c] = *reinterpret_cast<const u4*>(patch + r * X16_TR_ROW + (4 * c + g) * 16);
        __builtin_amdgcn_sched_barrier(0);
        first_ring(1);
        __syncthreads();                        // every wave has its fragments: the partial buffers may overwrite the patches
    }
    __builtin_amdgcn_sched_barrier(0);

    T* out = reinterpret_cast<T*>(a.out);
    if (a.out_step && out) out += (size_t)(*a.out_step) * a.out_step_stride;

    // epilogue coordinates: thread -> (tile of the trip, row, column)
    const int e_q = threadIdx.x >> 8, e_idx = threadIdx.x & 255, e_m = e_idx >> 4, e_nl = e_idx & 15;

    auto trip = [&](int it, auto pf_tag) {
        constexpr bool PF = decltype(pf_tag)::value;
        const int grp = (int)blockIdx.x + it * G, t0 = grp * X16_TPI;
        const int t_o = t0 + e_q, n = t_o * 16 + e_nl;
        float e_bias = 0.f;
        if (a.bias) e_bias = a.bias[min(n, a.N - 1)];
        v4f acc[X16_TPI];
#pragma unroll
        for (int q = 0; q < X16_TPI; ++q) {
            acc[q] = (v4f){0.f, 0.f, 0.f, 0.f};
            const u4* wn = tile_ptr(t0 + G * X16_TPI + q);
#pragma unroll
            for (int j = 0; j < X16_CPW; ++j) {
                acc[q] = mfma16(as_vec8<T>(ring[q * X16_CPW + j]), as_vec8<T>(xf[j]), acc[q]);
                if (PF) ring[q * X16_CPW + j] = ldg16_nt(wn + (unsigned)(j * 64 + lane));
                __builtin_amdgcn_sched_barrier(0);           // consume-j / refill-j order: the waits stay vmcnt(RING - 1)
            }
        }
        // D[n_local = 4 g + reg][m_local = r] -> red[buf][q][wave][m_local * 16 + n_local]
        float* rb = red + (size_t)(it & 1) * (X16_TPI * X16_WAVES * 256);
#pragma unroll
        for (int q = 0; q < X16_TPI; ++q)
            *reinterpret_cast<float4*>(&rb[(q * X16_WAVES + wa) * 256 + r * 16 + g * 4]) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < X16_WAVES; ++i) v += rb[(e_q * X16_WAVES + i) * 256 + e_idx];
        v += e_bias;
        const bool ok = (e_m < a.M) && (t_o < ntiles) && (n < a.N);
        if (EPI == EPI_NONE) {
            if (ok) out[(size_t)e_m * a.ldo + n] = fromf<T>(v);
        } else if (EPI == EPI_SILU_MUL) {
            // rows 0-7 of a tile are gate, 8-15 the matching up rows: the partner sits 8 lanes away in the same DPP row
            const float u = dpp_mov<DPP_ROR8>(v);
            if (a.out_packed) {
                // the 32-row fragment-packed block xrow16_k reads (row tile 0 of [f = k / 32][mt][lane (g = (k % 32) / 8, r = m)][8]): this
                // tile's 8 outputs k = 8 t_o .. + 8 of row m are one lane's 16-byte piece; rows >= M are zero-filled
                if (e_nl < 8 && t_o < ntiles)
                    out[((size_t)(((t_o >> 2) * 2) * 64 + (t_o & 3) * 16 + e_m) << 3) + e_nl] = e_m < a.M ? fromf<T>(swiglu<T>(v, u)) : fromf<T>(0.f);
            } else if (e_nl < 8 && ok) out[(size_t)e_m * a.ldo + t_o * 8 + e_nl] = fromf<T>(swiglu<T>(v, u));
        } else if (EPI == EPI_LOGITS) {
            float lv = rnd<T>(v);
            int li = n;
            const bool valid = n < a.n_valid && t_o < ntiles;
            if (valid && e_m < a.M && out) out[(size_t)e_m * a.ldo + n] = fromf<T>(lv);
            if (!valid) { lv = -INFINITY; li = 0x7fffffff; }
            // argmax over the tile's 16 columns (16 consecutive lanes share m); ties -> lowest index (torch.argmax)
#pragma unroll
            for (int sh = 8; sh > 0; sh >>= 1) {
                const float ov = __shfl_xor(lv, sh, 64);
                const int oi = __shfl_xor(li, sh, 64);
                if (ov > lv || (ov == lv && oi < li)) { lv = ov; li = oi; }
            }
            if (e_nl == 0 && e_m < a.M && t_o < ntiles) {
                a.part_val[(size_t)e_m * ntiles + t_o] = lv;
                a.part_idx[(size_t)e_m * ntiles + t_o] = li;
            }
        }
    };

    // the first trip is peeled: its waits cover the activation loads (newest in the queue), the loop's stay counted
    if (nit > 1) {
        trip(0, std::true_type{});
        for (int it = 1; it + 1 < nit; ++it) trip(it, std::true_type{});
    }
    trip(nit - 1, std::false_type{});
}

// ---- un-split o_proj / down_proj with the residual epilogue ---------------------------------------------------------------------------
// One workgroup per 16-column output tile, XR_WAVES waves split K evenly in whole 32-deep chunks. Per chunk a wave needs one KiB of weights
// (fragment-packed, HBM, read once: non-temporal) and one KiB of activations (row tile 0 of the producer's fragment-packed 32-row block: every
// workgroup reads the same block, L2-resident); both go through one register ring of XR_U chunk pairs refilled pair by pair as the MFMAs consume
// them. At 16 rows there is one tile per workgroup, so nothing would be gained by parking the activations anywhere.
constexpr int XR_WAVES = 16, XR_THREADS = XR_WAVES * 64, XR_U = 8;

template <typename T>
__global__ __launch_bounds__(XR_THREADS) void xrow16_k(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[XR_WAVES][256];
    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int KC = a.K >> 5;
    const int c0 = (KC * wa) / XR_WAVES, c1 = (KC * (wa + 1)) / XR_WAVES;       // wave-uniform; c1 - c0 >= 1 (K >= 512)
    const int clast = c1 - 1;
    const u4* wp = reinterpret_cast<const u4*>(a.W) + (size_t)tile * KC * 64 + lane;
    const T* X = reinterpret_cast<const T*>(a.X);
    // fragment (c, mt = 0) of the packed block. At <= 8 (<= 4) rows the lanes of the padding rows re-read row r - 8 (r & 3): their MFMA columns are never stored, and
    // the fragment's second 128-byte line per lane group (half of each line) is not fetched at all -- at K = 11008 the activations are as many L2 -> CU bytes as the weights
    const int xlane = a.M <= 4 ? (lane & ~12) : a.M <= 8 ? (lane & ~8) : (a.M <= 12 && (lane & 12) == 12) ? lane - 4 : lane;      // (9-12 rows: rows 12-15 re-read 8-11 -- half a line)
    auto xaddr = [&](int c) { return X + (size_t)(((c * 2) * 64 + xlane) * 8); };

    u4 wr[XR_U], xr[XR_U];
#pragma unroll
    for (int u = 0; u < XR_U; ++u) {
        const int c = min(c0 + u, clast);
        wr[u] = ldg16_nt(wp + (size_t)c * 64);
        xr[u] = ldg16(xaddr(c));
        __builtin_amdgcn_sched_barrier(0);
    }
    // the residual of this thread's output: requested behind the first ring, consumed in the epilogue
    const int e_m = (threadIdx.x & 255) >> 4, e_nl = threadIdx.x & 15, n = tile * 16 + e_nl;
    unsigned short res_bits = 0;
    if (threadIdx.x < 256) res_bits = reinterpret_cast<const unsigned short*>(a.resid)[(size_t)min(e_m, a.M - 1) * a.ldr + min(n, a.N - 1)];

    v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
    int cb = c0;
    for (; cb + XR_U < c1; cb += XR_U) {
#pragma unroll
        for (int u = 0; u < XR_U; ++u) {
            acc = mfma16(as_vec8<T>(wr[u]), as_vec8<T>(xr[u]), acc);
            const int c = min(cb + XR_U + u, clast);            // past the slice: re-read its last chunk (never multiplied)
            wr[u] = ldg16_nt(wp + (size_t)c * 64);
            xr[u] = ldg16(xaddr(c));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < XR_U; ++u)
        if (cb + u < c1) acc = mfma16(as_vec8<T>(wr[u]), as_vec8<T>(xr[u]), acc);      // wave-uniform guard
    *reinterpret_cast<float4*>(&red[wa][r * 16 + g * 4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (threadIdx.x < 256) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < XR_WAVES; ++i) v += red[i][threadIdx.x];
        if (e_m < a.M && n < a.N)
            reinterpret_cast<T*>(a.out)[(size_t)e_m * a.ldo + n] = fromf<T>(tof<T>(from_bits16<T>(res_bits)) + rnd<T>(v));
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
bool xs16_rows_ok(int M) { return M >= 3 && M <= 16; }

bool xstat16_supported(const GemmArgs& a, int epi) {
    return xs16_rows_ok(a.M) && a.K == X16_K && a.W && !a.W8 && (a.N + 15) / 16 >= 512 && a.norm_w && a.ldx % 8 == 0 && !a.xpacked &&
           (epi == EPI_NONE || epi == EPI_SILU_MUL || epi == EPI_LOGITS);
}

template <typename T, int EPI>
static void launch_xstat16_e(const GemmArgs& a, dim3 grid, hipStream_t s) {
    static DevOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute((const void*)xstat16_k<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X16_SMEM);
    hipLaunchKernelGGL((xstat16_k<T, EPI>), grid, dim3(X16_THREADS), X16_SMEM, s, a);          // always with the RMSNorm prologue (xstat16_supported)
}

void launch_xstat16(int dtype, const GemmArgs& a, int epi, hipStream_t s) {
    const int nt = (a.N + 15) / 16, groups = (nt + X16_TPI - 1) / X16_TPI;
    // 256 workgroups, unless the last round would be less than half full: then fewer workgroups with the same number of trips each (xstat32.hip)
    int g = groups < 256 ? groups : 256;
    const int rem = groups % 256;
    if (groups > 256 && rem && rem < 128) { const int trips = (groups + 255) / 256; g = (groups + trips - 1) / trips; }
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_NONE) launch_xstat16_e<T, EPI_NONE>(a, dim3(g), s);
        else if (epi == EPI_SILU_MUL) launch_xstat16_e<T, EPI_SILU_MUL>(a, dim3(g), s);
        else if (epi == EPI_LOGITS) launch_xstat16_e<T, EPI_LOGITS>(a, dim3(g), s);
    });
}

// X: the fragment-packed 32-row block (row tile 0 is read); resid / out row-major
bool xrow16_supported(const GemmArgs& a) {
    return xs16_rows_ok(a.M) && a.W && !a.W8 && a.K % 32 == 0 && a.K >= 32 * XR_WAVES && a.N % 16 == 0 && a.xpacked == 1 && a.resid && !a.bias && !a.norm_w;
}

void launch_xrow16(int dtype, const GemmArgs& a, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((xrow16_k<T>), dim3(a.N / 16), dim3(XR_THREADS), 0, s, a));
}

}  // namespace rdx

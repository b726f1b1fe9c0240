// Activation-stationary weight-streaming GEMM for 2 < M <= 32 rows and K = 4096 (batch 3-32 decode: QKV, gate/up, lm_head), gfx950.
//
// The first version of this path (skinny32_k, round 1; removed in round 3) re-staged the [32][K] activation block through LDS for
// every group of four output tiles, one workgroup barrier per 512-deep stage, and streamed weights at ~3.9 TB/s. Here the
// activations never move after start-up:
// 256 persistent 8-wave workgroups (one per CU), wave w owns the K range [512 w, 512 w + 512) and keeps the MFMA B-operand
// fragments of all 32 rows for that range in registers (2 x 16 fragments = 128 VGPRs, read once per workgroup = 256 KiB
// per CU from L2). The workgroup then walks output tiles t = blockIdx, blockIdx + grid, ...: per tile every wave streams its 16
// weight fragments (16 KiB contiguous, fragment-packed, non-temporal), does 32 MFMAs, and drops a 16 x 32 fp32 partial in LDS;
// one barrier per tile, after which all 512 threads reduce the 8 partials in a fixed order and run the epilogue while the
// next tile's fragments are already landing (the ring is refilled fragment by fragment as the MFMAs consume it: 16 KiB
// per wave = 128 KiB per CU in flight, no load ever waits for a barrier). fp8 weights (W8): 8 loads of 64-deep fragments per
// tile, two tiles per trip so the same 16 KiB stay in flight. Same rounding points / epilogues as the GEMV family (skinny_body.h).
#include <type_traits>
#include <algorithm>
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"   // swiglu()
#include "handoff.h"

namespace rdx {

constexpr int XS_WAVES = 8, XS_THREADS = 512, XS_K = 4096, XS_CPW = XS_K / 32 / XS_WAVES;   // 16 chunks of 32 per wave

// fp8 x fp8: one 16-byte piece of each operand = two v_mfma_f32_16x16x32_fp8_fp8 (bytes 0..7 and 8..15 of every lane, k = 64 c + 16 g + 8 h ..)
__device__ __forceinline__ v4f xs_mfma8(const u4& a, const u4& b, v4f c) {
    const long a0 = (long)(((unsigned long long)a.y << 32) | a.x), a1 = (long)(((unsigned long long)a.w << 32) | a.z);
    const long b0 = (long)(((unsigned long long)b.y << 32) | b.x), b1 = (long)(((unsigned long long)b.w << 32) | b.z);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
}

// A8 (with W8): the activations arrive as e4m3 bytes in the 64-deep fragment order (xpacked 4, written by rmsnorm4096_k<T, 4> with one scale per
// row): a piece is 16 bytes per lane like a weight piece, two fp8 MFMAs per pair, half the activation registers and half the start-up fetch;
// the fp32 sum is scaled by wscale[n] * xscale[m] in the epilogue.
// BLK (round 5: ONE prompt's prefill, M <= 192 rows): the rows are cut into blocks of 32 and every (tile walker, row block) pair is a workgroup of
// its own -- the NB row-block workgroups of a walker sit on the SAME XCD (workgroup ids are dealt round-robin over the 8 XCDs: id % 8) and walk the
// same tile sequence, so a weight fragment comes from HBM once and from that XCD's L2 NB - 1 times. X is the prompt's fragment-packed
// [k / 32][mtiles][lane][8] (xpacked 3, what rmsnorm_k<T, 3> / attention_k / the SwiGLU epilogue write); row block mb reads row tiles 2 mb, 2 mb + 1.
// Where a fragment load of row tile `mt` reads when only `m` of the block's 32 rows are real (round 5): a fragment is 16 rows x 16 bytes per lane group, i.e. two 128-byte lines
// of 8 rows each -- a lane whose row lies past the last line with a real row reads row `row % (8 ceil(m / 8))` instead (its MFMA column is never stored), so the lines of
// padding rows are never requested: at 12 rows half of every activation block (the 256 workgroups of a launch pull 32-64 MB of them through the L2s), at 40 rows 3/4 of
// the second row block. Returns the source lane (same lane group), sets the source row tile.
__device__ __forceinline__ int xs_src_lane(int m, int mt, int lane, int& mt_s) {
    const int m8 = (m + 7) & ~7, row = 16 * mt + (lane & 15);
    const int src = (m8 >= 32 || m8 <= 0) ? row : row % m8;
    mt_s = src >> 4;
    return (lane & 48) | (src & 15);
}

template <typename T, int EPI, bool W8, bool A8 = false, bool BLK = false>
__global__ __launch_bounds__(XS_THREADS) void xstat32_k(GemmArgs a) {
    static_assert(!A8 || W8, "fp8 activations go with fp8 weights");
    static_assert(!BLK || !W8 || A8, "row blocks with fp8 weights: fp8 x fp8 only");
    // (BLK with fp8, round 5: every 32-row block keeps the 32-row e4m3 / 64-deep layouts of the fp8 kernels, block b at a block stride -- xscale, the
    //  row-major outputs and the slabs are indexed by the global row)
    constexpr int TPI = W8 ? 2 : 1;                   // tiles per trip
    constexpr int LPT = W8 ? XS_CPW / 2 : XS_CPW;     // 16-byte weight loads per wave per tile
    constexpr int RING = TPI * LPT;                   // 16
    extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
    float* red = reinterpret_cast<float*>(smx);       // [2 bufs][TPI][8 waves][2 mt][256]

    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = (a.N + 15) >> 4;
    // walker id and walker count: plain mode = the grid; BLK: id = 8 slot + xcd, the 32 slots of an XCD are WPX walkers x NB row blocks
    int wid = (int)blockIdx.x, G = (int)gridDim.x, mb = 0;
    if (BLK) {
        const int NB = (a.mtiles + 1) >> 1, WPX = 32 / NB, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        if (slot >= WPX * NB) return;
        wid = xcd * WPX + slot / NB; mb = slot % NB; G = 8 * WPX;
    }
    const int ngroups = (ntiles + TPI - 1) / TPI;     // trips in total; this workgroup takes wid, wid + G, ...
    const int nit = (ngroups - wid + G - 1) / G;
    if (nit <= 0) return;
    // debug timeline (rdx_gemv_trace): 0 entry, 1 first trip's K loop done, 2 first trip done, 3 last trip begins, 4 its K loop done, 5 end
    long long* trc = (a.trace && threadIdx.x == 0) ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
#define XS_T(i) do { if (trc) trc[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    XS_T(0);

    const int WC = W8 ? XS_K / 64 : XS_K / 32;                       // weight chunks per tile row block
    const u4* wbase = reinterpret_cast<const u4*>(W8 ? a.W8 : a.W) + (size_t)(wa * LPT) * 64;   // wave-uniform
    auto tile_ptr = [&](int t) { return wbase + (size_t)min(t, ntiles - 1) * WC * 64; };

    u4 ring[RING];
    {
        const int t0 = wid * TPI;
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const u4* wp = tile_ptr(t0 + q);
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                ring[q * LPT + j] = ldg16_nt(wp + (unsigned)(j * 64 + lane));
                __builtin_amdgcn_sched_barrier(0);               // issue order = consume order (the loop's counted waits rely on it)
            }
        }
    }

    // ---- activations -> registers, requested AFTER the first ring of weights: the 256 workgroups read the same 256 KiB from L2
    // at once (~6 us of L2 hot-spotting), and weight requests queued behind them would leave HBM idle meanwhile. xf[mt][c] is the B fragment (column = row 16 mt + r of X, k = 8 g .. + 8 within the chunk)
    const T* X = reinterpret_cast<const T*>(a.X);
    constexpr int NXF = A8 ? XS_CPW / 2 : XS_CPW;
    u4 xf[2][NXF];
    const bool xp = a.xpacked != 0;          // fragment-packed by rmsnorm_k<T, PACK>: fragment (f, mt) is one contiguous KiB
    const int m_real = a.xdup_off ? 32 : min(32, a.M - 32 * mb);       // real rows of this 32-row block
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int mts;
        const int ls = xs_src_lane(m_real, mt, lane, mts);              // padding rows re-read real ones (xs_src_lane)
        if (A8) {
            // e4m3 block [chunk j][mt][lane][16]: this wave's chunks j = (XS_CPW / 2) wa .. + XS_CPW / 2
            const u4* x8 = reinterpret_cast<const u4*>(a.X) + (BLK ? (size_t)mb * (32 * XS_K / 16) : (size_t)0) + (size_t)((wa * (XS_CPW / 2)) * 2 + mts) * 64 + ls;
#pragma unroll
            for (int c = 0; c < NXF; ++c) xf[mt][c] = ldg16(x8 + (size_t)c * 128);
        } else if (BLK) {
            const int mtg = min(2 * mb + mts, a.mtiles - 1);
#pragma unroll
            for (int c = 0; c < NXF; ++c) xf[mt][c] = ldg16(X + ((size_t)(((wa * XS_CPW + c) * a.mtiles + mtg) * 64 + ls) << 3));
        } else {
            const T* xr = xp ? X + (size_t)(((wa * XS_CPW) * 2 + mts) * 64 + ls) * 8
                             : X + (size_t)min(mt * 16 + r, a.M - 1) * a.ldx + wa * (XS_CPW * 32);
#pragma unroll
            for (int c = 0; c < NXF; ++c) {
                // row-major X, bf16/f16 weights: chunk c covers k = 32 c + 8 g .. + 8. fp8: load j = c / 2 covers k = 64 j + 16 g .. + 16
                // and MFMA h = c & 1 takes k = 64 j + 16 g + 8 h .. + 8  (16 rows x 16 B per quarter wave: slow, tests only)
                const int koff = W8 ? ((c >> 1) * 64 + g * 16 + (c & 1) * 8) : (c * 32 + g * 8);
                xf[mt][c] = ldg16(xr + (xp ? c * 1024 : koff));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    T* out = reinterpret_cast<T*>(a.out);
    if (a.out_step && out) out += (size_t)(*a.out_step) * a.out_step_stride;

    // epilogue coordinates of this thread: one output per tile of the trip; idx = m_local * 16 + n_local
    const int e_mt = threadIdx.x >> 8, e_idx = threadIdx.x & 255, e_m = (BLK ? 32 * mb : 0) + e_mt * 16 + (e_idx >> 4), e_nl = e_idx & 15;

    auto trip = [&](int it, auto pf_tag) {
        constexpr bool PF = decltype(pf_tag)::value;
        const int grp = wid + it * G, t0 = grp * TPI;
        // operands the epilogue needs come first in the (in-order) load queue, ahead of the ring refills
        float e_res[TPI], e_sc[TPI], e_bias[TPI];
        const float e_xs = A8 ? a.xscale[e_m] : 1.f;                   // xscale[32]: one per row of the block (rows >= M: 1)
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const int n = min((t0 + q) * 16 + e_nl, ntiles * 16 - 1);
            e_res[q] = 0.f; e_sc[q] = 1.f; e_bias[q] = 0.f;
            if (EPI == EPI_RESID) e_res[q] = tof<T>(reinterpret_cast<const T*>(a.resid)[(size_t)min(e_m, a.M - 1) * a.ldr + min(n, a.N - 1)]);
            if (W8) e_sc[q] = a.wscale[n] * e_xs;
            if (a.bias) e_bias[q] = a.bias[min(n, a.N - 1)];
        }
        v4f acc[TPI][2];
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            acc[q][0] = (v4f){0.f, 0.f, 0.f, 0.f};
            acc[q][1] = (v4f){0.f, 0.f, 0.f, 0.f};
            const u4* wn = tile_ptr(t0 + G * TPI + q);
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const u4 wv = ring[q * LPT + j];
                if (A8) {
                    acc[q][0] = xs_mfma8(wv, xf[0][A8 ? j : 0], acc[q][0]);
                    acc[q][1] = xs_mfma8(wv, xf[1][A8 ? j : 0], acc[q][1]);
                } else if (W8) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u4 wd = dequant8<T>(h ? wv.z : wv.x, h ? wv.w : wv.y);
                        acc[q][0] = mfma16(as_vec8<T>(wd), as_vec8<T>(xf[0][A8 ? 0 : 2 * j + h]), acc[q][0]);
                        acc[q][1] = mfma16(as_vec8<T>(wd), as_vec8<T>(xf[1][A8 ? 0 : 2 * j + h]), acc[q][1]);
                    }
                } else {
                    acc[q][0] = mfma16(as_vec8<T>(wv), as_vec8<T>(xf[0][A8 ? 0 : j]), acc[q][0]);
                    acc[q][1] = mfma16(as_vec8<T>(wv), as_vec8<T>(xf[1][A8 ? 0 : j]), acc[q][1]);
                }
                if (PF) ring[q * LPT + j] = ldg16_nt(wn + (unsigned)(j * 64 + lane));
                __builtin_amdgcn_sched_barrier(0);           // keep consume-j / refill-j order: the waits stay vmcnt(RING - 1)
            }
        }
        if (it == 0) XS_T(1);
        if (!PF) XS_T(4);
        // D[n_local = 4 g + reg][m_local = r] -> red[buf][q][wave][mt][m_local * 16 + n_local]
        float* rb = red + (size_t)(it & 1) * (TPI * XS_WAVES * 2 * 256);
#pragma unroll
        for (int q = 0; q < TPI; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                *reinterpret_cast<float4*>(&rb[((q * XS_WAVES + wa) * 2 + mt) * 256 + r * 16 + g * 4]) =
                    make_float4(acc[q][mt][0], acc[q][mt][1], acc[q][mt][2], acc[q][mt][3]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            const int t_o = t0 + q, n = t_o * 16 + e_nl;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < XS_WAVES; ++i) v += rb[((q * XS_WAVES + i) * 2 + e_mt) * 256 + e_idx];
            if (W8) v *= e_sc[q];
            v += e_bias[q];
            const bool ok = (e_m < a.M) && (t_o < ntiles) && (n < a.N);
            if (EPI == EPI_NONE) {
                if (ok) out[(size_t)e_m * a.ldo + n] = fromf<T>(v);
            } else if (EPI == EPI_RESID) {
                if (ok) out[(size_t)e_m * a.ldo + n] = fromf<T>(e_res[q] + rnd<T>(v));
            } else if (EPI == EPI_SILU_MUL) {
                // rows 0-7 of a tile are gate, 8-15 the matching up rows: the partner sits 8 lanes away in the same DPP row
                const float u = dpp_mov<DPP_ROR8>(v);
                if (BLK && a.out_packed == 3) {
                    // the prompt's fragment-packed [k / 32][mtiles][lane][8] (out_packed 3: what wstat_k / this kernel's BLK mode read)
                    const int mtg = 2 * mb + e_mt;
                    if (e_nl < 8 && t_o < ntiles && mtg < a.mtiles)
                        out[((((size_t)(t_o >> 2) * a.mtiles + mtg) * 64 + (t_o & 3) * 16 + (e_idx >> 4)) << 3) + e_nl] =
                            e_m < a.M ? fromf<T>(swiglu<T>(v, u)) : fromf<T>(0.f);
                } else if (a.out_packed) {
                    // fragment-packed [f = k / 32][mt][lane (g = (k % 32) / 8, r = m % 16)][8] for the K-split consumer (xsplit32_k):
                    // this tile's 8 outputs k = 8 t_o .. + 8 of row m are one lane's 16-byte piece; rows >= M are zero-filled
                    // (out_packed 2, fp8 consumer: the 64-deep order -- piece t_o is fragment 2 (t_o / 8) + (t_o & 1), g = (t_o & 7) / 2)
                    const int pf = a.out_packed == 2 ? 2 * (t_o >> 3) + (t_o & 1) : (t_o >> 2), pg = a.out_packed == 2 ? ((t_o & 7) >> 1) : (t_o & 3);
                    if (e_nl < 8 && t_o < ntiles)
                        out[(BLK ? (size_t)mb * 32 * (ntiles * 8) : (size_t)0) + ((size_t)((pf * 2 + e_mt) * 64 + pg * 16 + (e_idx >> 4)) << 3) + e_nl] =
                            e_m < a.M ? fromf<T>(swiglu<T>(v, u)) : fromf<T>(0.f);
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
        }
    };

    // the first trip is peeled: its waits cover the activation loads (newest in the queue), the loop's stay counted
    if (nit > 1) {
        trip(0, std::true_type{});
        XS_T(2);
        for (int it = 1; it + 1 < nit; ++it) trip(it, std::true_type{});
    }
    XS_T(3);
    trip(nit - 1, std::false_type{});
    XS_T(5);
    if (trc) { trc[6] = nit; trc[7] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }    // HW_REG_XCC_ID
#undef XS_T
}

// ---- K-split variant for the 256-tile projections (down: K = 11008, o_proj: K = 4096) ------------------------------------------
// With N = 4096 there is one tile per CU, and [32][11008] activations do not fit one workgroup's registers. KGN workgroups share
// a tile: workgroup b is K group kg = b % KGN of tile slot b / KGN, its 8 waves hold the fragments of chunk range
// [KC s / S, KC (s + 1) / S), s = 8 kg + wave, S = 8 KGN (<= CPW chunks: 10-11 for K = 11008, KGN = 4), and it walks the tiles
// slot, slot + G / KGN, ... (4 trips of 88 KiB at G = 256). The activations must be fragment-packed (written by the producer's
// epilogue: xstat32_k's SwiGLU, or rmsnorm_k<T, 1>). Output: the fp32 partial of every K group in its own slab [kg][32][N],
// plain stores, no in-launch reduction -- the next kernel on the stream is always the RMSNorm of the following projection, and
// its prologue adds the slabs in kg order, rounds, adds the residual (rmsnorm_k with `slab`): the launch boundary is the sync.
// fp8 weights (W8): KC counts 64-deep chunks, one 16-byte load feeds two MFMAs per row tile, the activations are packed in the
// 64-deep order (PACK 2), the per-row scale is applied to the partial; two tiles per trip (TPI) keep 8-12 KiB per wave in flight.
// (A variant that ran that RMSNorm as a tail of this launch -- write-through slabs, arrival counter, workgroups 0..31 finishing one
// row each -- was measured 1.2 us slower per norm than the 5-us launch it replaced and made low-index workgroups wait on all others,
// against the liveness rule of handoff.h; removed in round 2.)
// A8 (with W8): the workgroup quantises ITS K range of the (model-dtype, fragment-packed) activations to e4m3 at start-up -- one absmax / 448
// scale per row over the range (cross-wave maximum through LDS) -- and multiplies fp8 x fp8; the partial carries wscale[n] * that scale. The
// ranges of the KGN groups are the `xgroups` K groups of the fp8 scheme (gemm8.hip; whole 128-deep blocks): o_proj 2, down_proj 4.
// BLK (round 5, 33-128 decoder rows): the rows in blocks of 32, the row-block workgroups of a (tile slot, K group) pair on one XCD like xstat32_k<.., BLK>;
// X is the fragment-packed [k / 32][mtiles][lane][8], the slabs are [KGN][16 mtiles][N].
template <typename T, int KC, int KGN, bool W8, int TPI, bool A8 = false, bool BLK = false>
__global__ __launch_bounds__(XS_THREADS) void xsplit32_k(GemmArgs a, float* __restrict__ slab) {
    static_assert(!A8 || W8, "fp8 activations go with fp8 weights");
    constexpr int SLOTS = XS_WAVES * KGN, CPW = (KC + SLOTS - 1) / SLOTS, FPL = W8 ? 2 : 1;   // fragments (MFMAs per row tile) per load
    extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
    float* red = reinterpret_cast<float*>(smx);       // [2 bufs][TPI][8 waves][2 mt][256]

    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = (a.N + 15) >> 4;
    int nts = (int)gridDim.x / KGN, kg = (int)blockIdx.x % KGN, ts = (int)blockIdx.x / KGN, mb = 0;
    if (BLK) {
        const int NB = (a.mtiles + 1) >> 1, PS = 32 / NB, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, p = slot / NB;
        if (slot >= PS * NB || p >= (PS / KGN) * KGN) return;
        mb = slot % NB; kg = p % KGN; nts = 8 * (PS / KGN); ts = xcd * (PS / KGN) + p / KGN;
    }
    if (ts >= nts) return;
    const int ntl = (ntiles - ts + nts - 1) / nts;    // tiles of this workgroup: ts, ts + nts, ...
    const int nit = (ntl + TPI - 1) / TPI;
    if (nit <= 0) return;
    long long* trc = (a.trace && threadIdx.x == 0) ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
#define XS_T(i) do { if (trc) trc[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    XS_T(0);
    const int slot = kg * XS_WAVES + wa;
    int c0 = (KC * slot) / SLOTS, cnt = (KC * (slot + 1)) / SLOTS - c0;      // wave-uniform
    if (A8) {
        // fp8 x fp8: the K range of group kg is the scheme's quantisation group -- whole 128-deep blocks [NB kg / KGN, NB (kg + 1) / KGN), NB = K / 128
        // (gemm8.hip) -- split evenly over the 8 waves (KC counts 64-deep chunks here)
        const int gs = 2 * (((KC >> 1) * kg) / KGN), ge = 2 * (((KC >> 1) * (kg + 1)) / KGN);
        c0 = gs + ((ge - gs) * wa) / XS_WAVES;
        cnt = gs + ((ge - gs) * (wa + 1)) / XS_WAVES - c0;
    }
    const u4* wbase = reinterpret_cast<const u4*>(W8 ? a.W8 : a.W) + (size_t)c0 * 64;
    auto tile_ptr = [&](int t) { return wbase + (size_t)min(t, ntiles - 1) * KC * 64; };

    u4 ring[TPI][CPW];
#pragma unroll
    for (int q = 0; q < TPI; ++q) {
        const u4* wp = tile_ptr(ts + q * nts);
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            ring[q][j] = ldg16_nt(wp + (unsigned)(min(j, cnt - 1) * 64 + lane));     // j >= cnt: re-read, multiplied by a zero fragment
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const T* X = reinterpret_cast<const T*>(a.X);
    u4 xf[2][CPW * FPL];
    const int m_real = a.xdup_off ? 32 : min(32, a.M - 32 * mb);       // real rows of this 32-row block: padding rows re-read real ones (xs_src_lane)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int mts;
        const int ls = xs_src_lane(m_real, mt, lane, mts);
#pragma unroll
        for (int j = 0; j < CPW * FPL; ++j) {
            const int f = (c0 + min(j / FPL, cnt - 1)) * FPL + (j % FPL);          // packed fragment index (32-deep, or 2 per 64-deep chunk)
            // BLK: the row-tile order of the model-dtype path (xpacked 3), or -- fp8 -- block mb's own 32-row block in the 64-deep order (xpacked 2)
            const u4 v = (BLK && !W8) ? ldg16(X + ((size_t)((f * a.mtiles + min(2 * mb + mts, a.mtiles - 1)) * 64 + ls) << 3))
                                      : ldg16(X + (BLK ? (size_t)mb * 32 * a.K : (size_t)0) + ((size_t)((f * 2 + mts) * 64 + ls) << 3));
            xf[mt][j] = (j / FPL) < cnt ? v : (u4){0u, 0u, 0u, 0u};
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    const int e_mt = threadIdx.x >> 8, e_idx = threadIdx.x & 255, e_m = (BLK ? 32 * mb : 0) + e_mt * 16 + (e_idx >> 4), e_nl = e_idx & 15;
    float* sl = slab + ((size_t)kg * (BLK ? (W8 ? 32 * ((a.mtiles + 1) >> 1) : 16 * a.mtiles) : 32) + e_m) * a.N;       // slab plane: the padded row count
    u4 xq[2][A8 ? CPW : 1];
    float e_xs = 1.f;
    if (A8) {
        // row maxima over this workgroup's K range: lane -> its 4 lane groups (same row, other k) -> the 8 waves through LDS (`red` is free here)
        float* rowmax = red;                       // [8 waves][32 rows]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float am = 0.f;
#pragma unroll
            for (int j = 0; j < CPW * FPL; ++j) am = fmaxf(am, amax8<T>(as_vec8<T>(xf[mt][j])));
            am = fmaxf(am, __shfl_xor(am, 16, 64));
            am = fmaxf(am, __shfl_xor(am, 32, 64));
            if (g == 0) rowmax[wa * 32 + mt * 16 + r] = am;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < XS_WAVES; ++i) am = fmaxf(am, rowmax[i * 32 + mt * 16 + r]);
            float sc, inv;
            fp8_scale(am, sc, inv);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const u2 lo = quant8<T>(as_vec8<T>(xf[mt][2 * j]), inv), hi = quant8<T>(as_vec8<T>(xf[mt][2 * j + 1]), inv);
                xq[mt][j] = (u4){lo.x, lo.y, hi.x, hi.y};
            }
        }
        {
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < XS_WAVES; ++i) am = fmaxf(am, rowmax[i * 32 + e_mt * 16 + (e_idx >> 4)]);       // this thread's row WITHIN the block
            float inv;
            fp8_scale(am, e_xs, inv);
        }
        __syncthreads();                           // rowmax is read before the first trip's partials overwrite `red`
    }

    auto trip = [&](int it, auto pf_tag) {
        constexpr bool PF = decltype(pf_tag)::value;
        float e_sc[TPI];
#pragma unroll
        for (int q = 0; q < TPI; ++q) e_sc[q] = W8 ? a.wscale[min((ts + (it * TPI + q) * nts) * 16 + e_nl, ntiles * 16 - 1)] : 1.f;
        v4f acc[TPI][2];
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            acc[q][0] = (v4f){0.f, 0.f, 0.f, 0.f};
            acc[q][1] = (v4f){0.f, 0.f, 0.f, 0.f};
            const u4* wn = tile_ptr(ts + ((it + 1) * TPI + q) * nts);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const u4 wv = ring[q][j];
                if (A8) {
                    acc[q][0] = xs_mfma8(wv, xq[0][A8 ? j : 0], acc[q][0]);
                    acc[q][1] = xs_mfma8(wv, xq[1][A8 ? j : 0], acc[q][1]);
                } else if (W8) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u4 wd = dequant8<T>(h ? wv.z : wv.x, h ? wv.w : wv.y);
                        acc[q][0] = mfma16(as_vec8<T>(wd), as_vec8<T>(xf[0][2 * j + h]), acc[q][0]);
                        acc[q][1] = mfma16(as_vec8<T>(wd), as_vec8<T>(xf[1][2 * j + h]), acc[q][1]);
                    }
                } else {
                    acc[q][0] = mfma16(as_vec8<T>(wv), as_vec8<T>(xf[0][j]), acc[q][0]);
                    acc[q][1] = mfma16(as_vec8<T>(wv), as_vec8<T>(xf[1][j]), acc[q][1]);
                }
                if (PF) ring[q][j] = ldg16_nt(wn + (unsigned)(min(j, cnt - 1) * 64 + lane));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (it == 0) XS_T(1);
        if (!PF) XS_T(4);
        float* rb = red + (size_t)(it & 1) * (TPI * XS_WAVES * 2 * 256);
#pragma unroll
        for (int q = 0; q < TPI; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                *reinterpret_cast<float4*>(&rb[((q * XS_WAVES + wa) * 2 + mt) * 256 + r * 16 + g * 4]) =
                    make_float4(acc[q][mt][0], acc[q][mt][1], acc[q][mt][2], acc[q][mt][3]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TPI; ++q) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < XS_WAVES; ++i) v += rb[((q * XS_WAVES + i) * 2 + e_mt) * 256 + e_idx];
            const int tl = it * TPI + q, n = (ts + tl * nts) * 16 + e_nl;
            if (tl < ntl && e_m < a.M && n < a.N) sl[n] = v * e_sc[q] * e_xs;
        }
    };
    if (nit > 1) {
        trip(0, std::true_type{});
        XS_T(2);
        for (int it = 1; it + 1 < nit; ++it) trip(it, std::true_type{});
    }
    XS_T(3);
    trip(nit - 1, std::false_type{});
    XS_T(5);
    if (trc) { trc[6] = nit; trc[7] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
#undef XS_T
}

// Round 6: the K-split projections of the fp8 configuration's DECODE (o_proj, down_proj at 3-128 rows) multiply W8A16 -- the e4m3 weight fragments expanded to the model
// dtype in registers (exact), the producer's model-dtype activation fragments as they are. Their fp8 x fp8 form (rounds 3-5; RDX_FP8_SPLIT_A8=1: the A/B leg) had every one
// of the 256 workgroups load its K range of the model-dtype activations anyway, reduce the row maxima across its waves through LDS (two barriers) and convert -- ~3 us of a
// 9.6-us o_proj whose weight stream is 16.8 MB -- and it is the less accurate of the two (profiles/r06_fp8_variants.md). The prefill and the activation-stationary
// projections (QKV, gate/up, lm_head: their e4m3 rows come out of the RMSNorm launch for free) stay fp8 x fp8.
static bool fp8_split_a8() {
    static const bool a8 = getenv("RDX_FP8_SPLIT_A8") && atoi(getenv("RDX_FP8_SPLIT_A8")) == 1;
    return a8;
}

// RDX_XDUP=0 (A/B leg): the padding rows of a 32-row block load their own lines
static GemmArgs xs_env(GemmArgs a) {
    static const bool off = getenv("RDX_XDUP") && atoi(getenv("RDX_XDUP")) == 0;
    a.xdup_off = off ? 1 : 0;
    return a;
}

// smallest row count (batch) that takes the activation-stationary / K-split kernels; below it the GEMV family of skinny_body.h
// (whose LDS stage holds up to 4 rows of 4096). Measured step times, default GEMV path -> these kernels: batch 5 3.31 -> 3.25 ms,
// batch 8 3.53 -> 3.29, batch 16 4.23 -> 3.53 (rows beyond the batch ride along as zero columns of the 32-row block); batch 3 and
// 4, whose rows would still fit the GEMV's LDS stage: 3.39 / 3.56 ms there. Batch 1-2 keep the fused / chained GEMV launches.
int xs_min_rows() {
    return 3;
}

// K groups for this shape (0 = not supported): needs fragment-packed activations (xpacked 1; 2 = the fp8 64-deep order)
int xsplit32_groups(const GemmArgs& a) {
    const bool w8 = a.W8 && a.wscale;
    if (!(a.M >= xs_min_rows() && a.M <= 32) || a.xpacked != (w8 ? 2 : 1) || a.norm_w || a.bias || (a.N + 15) / 16 < 128 || (a.N + 15) / 16 > 512) return 0;
    if (a.K == 11008) return 4;
    if (a.K == 4096) return 2;
    return 0;
}

void launch_xsplit32(int dtype, const GemmArgs& a_in, float* slab, hipStream_t s) {
    const GemmArgs a = xs_env(a_in);
    const int kgn = xsplit32_groups(a);
    const int nt = (a.N + 15) / 16;
    const bool w8 = a.W8 && a.wscale;                        // fp8 weights: fp8 x fp8, every K-group workgroup quantises its range of the activations
    const size_t smem = (size_t)2 * (w8 ? 2 : 1) * XS_WAVES * 2 * 256 * 4;
    RDX_DISPATCH_T(dtype, T, {
        if (kgn == 4) {
            const int nts = std::min(nt, 256 / 4);
            if (w8 && fp8_split_a8()) hipLaunchKernelGGL((xsplit32_k<T, 172, 4, true, 2, true>), dim3(nts * 4), dim3(XS_THREADS), smem, s, a, slab);
            else if (w8) hipLaunchKernelGGL((xsplit32_k<T, 172, 4, true, 2, false>), dim3(nts * 4), dim3(XS_THREADS), smem, s, a, slab);
            else hipLaunchKernelGGL((xsplit32_k<T, 344, 4, false, 1>), dim3(nts * 4), dim3(XS_THREADS), smem, s, a, slab);
        } else if (kgn == 2) {
            const int nts = std::min(nt, 256 / 2);
            if (w8 && fp8_split_a8()) hipLaunchKernelGGL((xsplit32_k<T, 64, 2, true, 2, true>), dim3(nts * 2), dim3(XS_THREADS), smem, s, a, slab);
            else if (w8) hipLaunchKernelGGL((xsplit32_k<T, 64, 2, true, 2, false>), dim3(nts * 2), dim3(XS_THREADS), smem, s, a, slab);
            else hipLaunchKernelGGL((xsplit32_k<T, 128, 2, false, 1>), dim3(nts * 2), dim3(XS_THREADS), smem, s, a, slab);
        }
    });
}

// fp8 x fp8 on row blocks (33-128 rows, round 5): X = per-block 32-row layouts at a block stride -- xstat: e4m3 blocks (xpacked 4) + xscale[rows];
// xsplit: model-dtype 64-deep blocks (xpacked 2), each K-group workgroup quantises its range; slabs [groups][32 NB][N]
bool xstat_blk8_supported(const GemmArgs& a, int epi) {
    return a.xpacked == 4 && a.xscale && a.mtiles >= 3 && a.mtiles <= 8 && a.M <= a.mtiles * 16 && a.K == XS_K && a.W8 && a.wscale && !a.norm_w && !a.bias &&
           (epi == EPI_NONE || epi == EPI_SILU_MUL || epi == EPI_LOGITS) && (a.out_packed == 0 || (a.out_packed == 2 && epi == EPI_SILU_MUL)) && (a.N + 15) / 16 >= 128;
}

void launch_xstat_blk8(int dtype, const GemmArgs& a_in, int epi, hipStream_t s) {
    const GemmArgs a = xs_env(a_in);
    const size_t smem = (size_t)2 * 2 * XS_WAVES * 2 * 256 * 4;
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_NONE) hipLaunchKernelGGL((xstat32_k<T, EPI_NONE, true, true, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
        else if (epi == EPI_SILU_MUL) hipLaunchKernelGGL((xstat32_k<T, EPI_SILU_MUL, true, true, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
        else if (epi == EPI_LOGITS) hipLaunchKernelGGL((xstat32_k<T, EPI_LOGITS, true, true, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
    });
}

int xsplit_blk8_groups(const GemmArgs& a) {
    if (a.xpacked != 2 || !a.W8 || !a.wscale || a.mtiles < 3 || a.mtiles > 8 || a.M > a.mtiles * 16 || a.norm_w || a.bias || (a.N + 15) / 16 < 128 || (a.N + 15) / 16 > 512) return 0;
    return a.K == 11008 ? 4 : a.K == 4096 ? 2 : 0;
}

void launch_xsplit_blk8(int dtype, const GemmArgs& a_in, float* slab, hipStream_t s) {
    const GemmArgs a = xs_env(a_in);
    const size_t smem = (size_t)2 * 2 * XS_WAVES * 2 * 256 * 4;
    RDX_DISPATCH_T(dtype, T, {
        if (fp8_split_a8()) {
            if (a.K == 11008) hipLaunchKernelGGL((xsplit32_k<T, 172, 4, true, 2, true, true>), dim3(256), dim3(XS_THREADS), smem, s, a, slab);
            else hipLaunchKernelGGL((xsplit32_k<T, 64, 2, true, 2, true, true>), dim3(256), dim3(XS_THREADS), smem, s, a, slab);
        } else {
            if (a.K == 11008) hipLaunchKernelGGL((xsplit32_k<T, 172, 4, true, 2, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a, slab);
            else hipLaunchKernelGGL((xsplit32_k<T, 64, 2, true, 2, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a, slab);
        }
    });
}

bool xsplit_blk_supported(const GemmArgs& a) {
    return a.xpacked == 3 && a.mtiles >= 3 && a.mtiles <= 12 && a.M <= a.mtiles * 16 && a.M > (a.mtiles - 1) * 16 && a.K == 11008 && a.W && !a.W8 && !a.norm_w && !a.bias &&
           (a.N + 15) / 16 >= 128 && (a.N + 15) / 16 <= 512;
}

void launch_xsplit_blk(int dtype, const GemmArgs& a_in, float* slab, hipStream_t s) {
    const GemmArgs a = xs_env(a_in);
    const size_t smem = (size_t)2 * XS_WAVES * 2 * 256 * 4;
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((xsplit32_k<T, 344, 4, false, 1, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a, slab));
}

bool xstat32_supported(const GemmArgs& a, int epi) {
    constexpr int min_tiles = 512;                            // fewer tiles: the K-split variant (256-tile projections) or the GEMV family
    return a.M >= xs_min_rows() && a.M <= 32 && a.K == XS_K && !a.norm_w && (a.N + 15) / 16 >= min_tiles &&
           (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_SILU_MUL || epi == EPI_LOGITS);
}

template <typename T, bool W8, bool A8 = false>
static void launch_xstat32_t(const GemmArgs& a, int epi, hipStream_t s) {
    constexpr int TPI = W8 ? 2 : 1;
    const int nt = (a.N + 15) / 16, groups = (nt + TPI - 1) / TPI;
    // 256 workgroups, unless the last round would be less than half full: then fewer workgroups with the same number of trips each
    // (QKV: 769 tiles = 3 rounds + 1 tile -> 193 workgroups x 4 trips, 20.2 -> 19.5 us; gate/up 1376 -> 230 x 6, 32.4 -> 31.9 us;
    // lm_head's 2001 tiles fill 82 % of their last round and lose 0.9 us that way)
    int g = groups < 256 ? groups : 256;
    const int rem = groups % 256;
    if (groups > 256 && rem && rem < 128) { const int trips = (groups + 255) / 256; g = (groups + trips - 1) / trips; }
    dim3 grid(g), block(XS_THREADS);
    const size_t smem = (size_t)2 * TPI * XS_WAVES * 2 * 256 * 4;
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((xstat32_k<T, EPI_NONE, W8, A8>), grid, block, smem, s, a); break;
        case EPI_RESID: hipLaunchKernelGGL((xstat32_k<T, EPI_RESID, W8, A8>), grid, block, smem, s, a); break;
        case EPI_SILU_MUL: hipLaunchKernelGGL((xstat32_k<T, EPI_SILU_MUL, W8, A8>), grid, block, smem, s, a); break;
        case EPI_LOGITS: hipLaunchKernelGGL((xstat32_k<T, EPI_LOGITS, W8, A8>), grid, block, smem, s, a); break;
        default: break;
    }
}

// ONE prompt's projections (K = 4096) in row blocks of 32: see BLK above. X fragment-packed (xpacked 3, a.mtiles row tiles), epilogues NONE / RESID /
// SILU_MUL (out_packed 3: the same packed order for the next projection). Every XCD runs 32 / NB walkers with NB = ceil(mtiles / 2) row blocks each.
bool xstat_blk_supported(const GemmArgs& a, int epi) {
    return a.xpacked == 3 && a.mtiles >= 1 && a.mtiles <= 12 && a.M <= a.mtiles * 16 && a.M > (a.mtiles - 1) * 16 && a.K == XS_K && a.W && !a.W8 && !a.norm_w &&
           !a.bias && (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_SILU_MUL || epi == EPI_LOGITS) && (a.out_packed == 0 || (a.out_packed == 3 && epi == EPI_SILU_MUL)) &&
           (a.N + 15) / 16 >= 128;
}

void launch_xstat_blk(int dtype, const GemmArgs& a_in, int epi, hipStream_t s) {
    const GemmArgs a = xs_env(a_in);
    const size_t smem = (size_t)2 * XS_WAVES * 2 * 256 * 4;
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_NONE) hipLaunchKernelGGL((xstat32_k<T, EPI_NONE, false, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
        else if (epi == EPI_RESID) hipLaunchKernelGGL((xstat32_k<T, EPI_RESID, false, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
        else if (epi == EPI_SILU_MUL) hipLaunchKernelGGL((xstat32_k<T, EPI_SILU_MUL, false, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
        else if (epi == EPI_LOGITS) hipLaunchKernelGGL((xstat32_k<T, EPI_LOGITS, false, false, true>), dim3(256), dim3(XS_THREADS), smem, s, a);
    });
}

void launch_xstat32(int dtype, const GemmArgs& a_in, int epi, hipStream_t s) {
    const GemmArgs a = xs_env(a_in);
    const bool w8 = a.W8 && a.wscale;        // fp8 weights: the activations are the e4m3 block of rmsnorm4096_k<T, 4> (xpacked 4; launch_skinny_gemm checks)
    RDX_DISPATCH_T(dtype, T, {
        if (w8) launch_xstat32_t<T, true, true>(a, epi, s);
        else launch_xstat32_t<T, false>(a, epi, s);
    });
}

}  // namespace rdx

// Weight-streaming GEMM for 16 < M <= 32 activation rows (batch-32 decode), gfx950.
//
// The batch-1 GEMV (skinny_body.h) stages the whole activation block in LDS; at M = 32, K = 4096 that is 256 KiB and does not
// fit, and streaming the activation fragments from L2 per wave costs 2x the weight bytes per tile (352 MB of L2 traffic for
// the 180 MB gate/up weights: the kernel ran at half the HBM rate). Here a 16-wave workgroup owns SUB output tiles
// (16 columns each) with WPS = 16/SUB waves splitting K per tile, exactly like the batch-1 kernel, but the activations
// are staged through LDS in K stages shared by all the workgroup's tiles: per stage every wave consumes CH chunks of 32
// (its own K range), so the stage buffer holds [WPS ranges][CH chunks][32 rows][32] elements = 32 KiB at WPS = CH = 4,
// double-buffered; the activation traffic drops to (rows x K x 2 B) per SUB tiles. Weights: fragment-packed (gemm.hip), a ring of
// 8 chunks per wave in registers (unconditional clamped loads, counted waits). The LDS image of a chunk is [row][32] so a wave's
// MFMA B-operand read is one contiguous KiB. Fixed-order LDS reduction over the WPS partial tiles, fused epilogues as in
// skinny_body.h (same rounding points).
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"   // swiglu()

namespace rdx {

constexpr int S32_WAVES = 16, S32_THREADS = 1024;

// W8: fp8 (e4m3) weights, 64-deep fragment order + per-row scale (gemm.hip pack_weight_fp8_k): a wave's stage is then CH/2
// 16-byte loads (each feeds two MFMAs), the ring is twice as many stages deep so that 8 KiB per wave stay in flight.
template <typename T, int EPI, int SUB, bool W8>
__global__ __launch_bounds__(S32_THREADS, 8) void skinny32_k(GemmArgs a) {
    typedef typename Vec8<T>::type V8;
    constexpr int WPS = S32_WAVES / SUB, CH = 16 / WPS, MT = 2;   // WPS * CH = 16 chunk slots per stage: a 32 KiB stage image
    constexpr int WL = W8 ? CH / 2 : CH;                           // weight loads (16 B per lane) per wave per stage
    constexpr int NWS = W8 ? 2 : 8 / WL;                           // weight ring: 8 loads (8 KiB) per wave in flight; fp8: 4 loads = the
                                                                   // same K depth (the dequantisation temporaries need the registers)
    static_assert(!W8 || CH % 2 == 0, "fp8: a stage must hold whole 64-deep chunks");
    constexpr int STAGE_U4 = WPS * CH * 32 * 4;                  // 16-byte pieces per stage buffer
    constexpr int XPT = STAGE_U4 / S32_THREADS;                  // pieces per thread per stage
    static_assert(STAGE_U4 % S32_THREADS == 0, "stage must tile over the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm32[];
    u4* xbuf = reinterpret_cast<u4*>(sm32);                      // [2][WPS][CH][32 rows][4 pieces]
    float* red = reinterpret_cast<float*>(sm32);                 // after the K loop: [16 waves][MT][256]

    const int lane = threadIdx.x & 63, wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = wa / WPS, w = wa - sub * WPS;
    const int ntiles = (a.N + 15) >> 4;
    const int tile = blockIdx.x * SUB + sub, tile_c = min(tile, ntiles - 1);
    const int r = lane & 15, g = lane >> 4;
    const int K = a.K, KC = K >> 5;
    const int nst = KC / (WPS * CH);                             // stages: every wave consumes CH chunks of 32 per stage
    const int WC = W8 ? (KC >> 1) : KC;                          // weight chunks per row (64-deep for fp8)
    const int c0 = (WC * w) / WPS, c1 = (WC * (w + 1)) / WPS;    // this wave's weight-chunk range
    const T* X = reinterpret_cast<const T*>(a.X);
    const u4* wtile = reinterpret_cast<const u4*>(W8 ? a.W8 : a.W) + (size_t)tile_c * WC * 64;   // wave-uniform
    const int clast = min(max(c1 - 1, c0), WC - 1);

    // activation piece idx -> (range q, chunk j, row m, piece p) of the stage image. K is a multiple of 16 chunks
    // (skinny32_supported), so every range is a whole number of stages and a piece's address advances by CH chunks per
    // stage: one base pointer per piece. Rows >= M read row M-1 (finite; their outputs are never stored).
    unsigned xoff[XPT];                                            // 32-bit element offsets: scalar base + VGPR offset addressing
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = threadIdx.x + i * S32_THREADS;
        const int xp = idx & 3, xm = (idx >> 2) & 31, xj = (idx >> 7) % CH, xq = (idx >> 7) / CH;
        xoff[i] = (unsigned)(min(xm, a.M - 1) * a.ldx + (xq * (KC / WPS) + xj) * 32 + xp * 8);
    }
    auto load_x = [&](u4 (&dst)[XPT], int s) {
        const unsigned so = (unsigned)(min(s, nst - 1) * (CH * 32));  // stages past the end re-read the last one (unused)
#pragma unroll
        for (int i = 0; i < XPT; ++i) dst[i] = ldg16(X + (size_t)(xoff[i] + so));
    };
    auto store_x = [&](const u4 (&src)[XPT], int buf) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) xbuf[buf * STAGE_U4 + threadIdx.x + i * S32_THREADS] = src[i];
    };
    auto load_w = [&](u4 (&dst)[WL], int s) {
#pragma unroll
        for (int j = 0; j < WL; ++j) dst[j] = ldg16_nt(wtile + (size_t)(unsigned)(min(c0 + s * WL + j, clast) * 64 + lane));
    };

    v4f acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u4 (&wr)[WL], int s, int buf) {
        // stage image: piece index ((q*CH + j)*32 + m)*4 + p ; this wave: q = w, lane (g, r) reads row 16*mt + r, piece g
        const u4* xb = xbuf + buf * STAGE_U4 + (size_t)(w * CH) * 128;
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            if (W8) {
                // 64-deep chunk j: lane (g, r) holds k = 64j + 16g .. +16; MFMA h takes k = 64j + 16g + 8h .. +8, which sits in
                // the image's 32-deep chunk 2j + (g >> 1), piece (2g + h) & 3
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u4 wd = dequant8<T>(h ? wr[j].z : wr[j].x, h ? wr[j].w : wr[j].y);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const u4 xv = xb[((2 * j + (g >> 1)) * 32 + mt * 16 + r) * 4 + ((2 * g + h) & 3)];
                        acc[mt] = mfma16(as_vec8<T>(wd), as_vec8<T>(xv), acc[mt]);
                    }
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u4 xv = xb[(j * 32 + mt * 16 + r) * 4 + g];
                    acc[mt] = mfma16(as_vec8<T>(wr[j]), as_vec8<T>(xv), acc[mt]);
                }
            }
        }
    };

    // Weight ring NWS stages deep, activation image one stage ahead. Within a stage the activation loads are issued
    // BEFORE the weight loads: vmcnt is in-order, so the wait in front of the LDS store then leaves this stage's weight
    // loads in flight (the other order drains the ring every stage). Every load is unconditional (addresses clamped) ->
    // counted waits; stages past the end compute nothing (guards in compute()). Register budget: 64 VGPRs (two workgroups
    // per CU) -- a spill reload inside the loop would force vmcnt(0).
    u4 wr[NWS][WL], xr[XPT];
#pragma unroll
    for (int k = 0; k + 1 < NWS; ++k) load_w(wr[k], k);
    load_x(xr, 0);
    store_x(xr, 0);
    __syncthreads();
    for (int s0 = 0; s0 < nst; s0 += NWS) {
#pragma unroll
        for (int k = 0; k < NWS; ++k) {
            const int s = s0 + k;
            load_x(xr, s + 1);
            load_w(wr[(k + NWS - 1) % NWS], s + NWS - 1);
            if (s < nst) compute(wr[k], s, s & 1);                  // workgroup-uniform (MFMA ignores EXEC)
            store_x(xr, (s + 1) & 1);
            __syncthreads();
        }
    }

    // D[n_local = g*4+reg][m_local = r] -> red[wave][mt][m_local*16 + n_local]   (the stage buffers are free: barrier above)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<float4*>(&red[(wa * MT + mt) * 256 + r * 16 + g * 4]) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    __syncthreads();

    T* out = reinterpret_cast<T*>(a.out);
    if (a.out_step && out) out += (size_t)(*a.out_step) * a.out_step_stride;
    for (int o = threadIdx.x; o < SUB * MT * 256; o += S32_THREADS) {
        const int so = o / (MT * 256), mt = (o >> 8) % MT, idx = o & 255, m_local = idx >> 4, n_local = idx & 15;
        const int t_o = blockIdx.x * SUB + so;
        const int m = mt * 16 + m_local, n = t_o * 16 + n_local;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < WPS; ++i) v += red[((so * WPS + i) * MT + mt) * 256 + idx];
        if (W8) v *= a.wscale[min(n, ntiles * 16 - 1)];
        if (a.bias && n < a.N) v += a.bias[n];
        const bool ok = (m < a.M) && (t_o < ntiles) && (n < a.N);
        if (EPI == EPI_NONE) {
            if (ok) out[(size_t)m * a.ldo + n] = fromf<T>(v);
        } else if (EPI == EPI_RESID) {
            if (ok) {
                const float rsd = tof<T>(reinterpret_cast<const T*>(a.resid)[(size_t)m * a.ldr + n]);
                out[(size_t)m * a.ldo + n] = fromf<T>(rsd + rnd<T>(v));
            }
        } else if (EPI == EPI_SILU_MUL) {
            float u = 0.f;
#pragma unroll
            for (int i = 0; i < WPS; ++i) u += red[((so * WPS + i) * MT + mt) * 256 + ((idx + 8) & 255)];
            if (W8) u *= a.wscale[min(t_o, ntiles - 1) * 16 + ((n_local + 8) & 15)];
            if (n_local < 8 && ok) out[(size_t)m * a.ldo + t_o * 8 + n_local] = fromf<T>(swiglu<T>(v, u));
        } else if (EPI == EPI_LOGITS) {
            float lv = rnd<T>(v);
            int li = n;
            const bool valid = n < a.n_valid && t_o < ntiles;
            if (valid && m < a.M && out) out[(size_t)m * a.ldo + n] = fromf<T>(lv);
            if (!valid) { lv = -INFINITY; li = 0x7fffffff; }
            // argmax over the tile's 16 columns (16 consecutive lanes share m); ties -> lowest index (torch.argmax)
#pragma unroll
            for (int sh = 8; sh > 0; sh >>= 1) {
                const float ov = __shfl_xor(lv, sh, 64);
                const int oi = __shfl_xor(li, sh, 64);
                if (ov > lv || (ov == lv && oi < li)) { lv = ov; li = oi; }
            }
            if (n_local == 0 && m < a.M && t_o < ntiles) {
                a.part_val[(size_t)m * ntiles + t_o] = lv;
                a.part_idx[(size_t)m * ntiles + t_o] = li;
            }
        }
    }
}

bool skinny32_supported(const GemmArgs& a, int epi) {
    // measured at M = 32: 4-tile workgroups win for the many-tile GEMMs (gate/up 69 -> 55 us, QKV 46 -> 33, lm_head 92 -> 65);
    // with <= 512 tiles the 2-tile variant loses to the L2-streaming kernel (o_proj 12.8 -> 19 us, down 31 -> 48)
    return a.M > 16 && a.M <= 32 && a.K % 512 == 0 && !a.norm_w && (a.N + 15) / 16 > 512 &&
           (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_SILU_MUL || epi == EPI_LOGITS);
}

template <typename T, int SUB, bool W8>
static void launch_skinny32_sub(const GemmArgs& a, int epi, hipStream_t s) {
    const int nt = (a.N + 15) / 16;
    dim3 grid((nt + SUB - 1) / SUB), block(S32_THREADS);
    const size_t stage = (size_t)2 * 16 * 32 * 4 * 16, redb = (size_t)S32_WAVES * 2 * 256 * 4;   // 2 x 32 KiB stage images
    const size_t smem = stage > redb ? stage : redb;
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((skinny32_k<T, EPI_NONE, SUB, W8>), grid, block, smem, s, a); break;
        case EPI_RESID: hipLaunchKernelGGL((skinny32_k<T, EPI_RESID, SUB, W8>), grid, block, smem, s, a); break;
        case EPI_SILU_MUL: hipLaunchKernelGGL((skinny32_k<T, EPI_SILU_MUL, SUB, W8>), grid, block, smem, s, a); break;
        case EPI_LOGITS: hipLaunchKernelGGL((skinny32_k<T, EPI_LOGITS, SUB, W8>), grid, block, smem, s, a); break;
        default: break;
    }
}

void launch_skinny32(int dtype, const GemmArgs& a, int epi, hipStream_t s) {
    // 4 tiles x 4 waves per workgroup (2 tiles x 8 waves measured slower at every shape of the decode step)
    const bool w8 = a.W8 && a.wscale;                         // K % 512 == 0 holds (skinny32_supported): whole 64-deep chunks
    RDX_DISPATCH_T(dtype, T, { if (w8) launch_skinny32_sub<T, 4, true>(a, epi, s); else launch_skinny32_sub<T, 4, false>(a, epi, s); });
}

}  // namespace rdx

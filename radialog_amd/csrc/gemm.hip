// MFMA GEMM kernels over fragment-packed weights (gfx950, mfma_f32_16x16x32_{f16,bf16}).
//
// Every dense contraction on the RaDialog hot path is  out[M,N] = X[M,K] . W[N,K]^T (+ epilogue)  with W a fixed
// model weight, so weights are re-laid-out ONCE at load time into the MFMA operand order
//     Wp[n_tile16][k_chunk32][lane 0..63][8 elems],  lane = (g<<4)|r  holds  W[n_tile*16+r][k_chunk*32+g*8 .. +8]
// -> one wave-wide 16-byte load is a fully coalesced 1 KiB read that lands directly in the MFMA A operand.
// Activations are the MFMA B operand, so D[i][j] = out[m = j][n = i]: lane (r = m_local, g) ends up with 4 CONSECUTIVE
// output columns n = g*4 .. g*4+3 of row m -> 8-byte stores, no LDS transpose.
//
//   skinny_gemm : M <= 32 (single-token decode at batch 1..32, Q-Former at batch 1). HBM-bound weight streaming:
//                 one workgroup per 16 output columns, its 8 waves split K, partial tiles reduced through LDS,
//                 fused RMSNorm prologue and fused bias / residual / SwiGLU / logits+argmax epilogues.
//                 Replaces the reference's nn.Linear GEMVs in LlamaAttention / LlamaMLP / lm_head
//                 (modeling_llama_imgemb.py:158-159,:198-200,:245,:768) and LlamaRMSNorm (:85-93).
//   tiled_gemm  : M > 32 (prefill, ResNet convs as implicit GEMM, Q-Former at batch > 1). 128x128 block tile,
//                 2x2 waves x (4x4) MFMA tiles, activations staged through LDS in fragment order (conflict-free
//                 ds_read_b128), weights straight from global in packed order. Conv mode gathers the im2col row
//                 on the fly from an NHWC tensor (torchvision Bottleneck convs behind biovil_t/resnet.py:34-42).
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"

namespace rdx {

// ------------------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_k(const float* __restrict__ src, T* __restrict__ dst, int N, int K, int Npad,
                              const int* __restrict__ rowmap) {
    const int K8 = K >> 3;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Npad * K8) return;
    const int i = (int)(idx / K8), k8 = (int)(idx % K8);
    int srow = rowmap ? rowmap[i] : i;
    if (srow >= N) srow = -1;
    typename Vec8<T>::type v;
    if (srow >= 0) {
        const float4* s = reinterpret_cast<const float4*>(src + (size_t)srow * K + (size_t)k8 * 8);
        float4 a = s[0], b = s[1];
        v[0] = fromf<T>(a.x); v[1] = fromf<T>(a.y); v[2] = fromf<T>(a.z); v[3] = fromf<T>(a.w);
        v[4] = fromf<T>(b.x); v[5] = fromf<T>(b.y); v[6] = fromf<T>(b.z); v[7] = fromf<T>(b.w);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fromf<T>(0.f);
    }
    const int KC = K >> 5;
    const int nt = i >> 4, r = i & 15, kc = k8 >> 2, g = k8 & 3;
    u4* d = reinterpret_cast<u4*>(dst) + ((size_t)nt * KC + kc) * 64 + (g * 16 + r);
    *d = as_u4<T>(v);
}

void launch_pack_weight(int dtype, const float* src, void* dst, int N, int K, int Npad, const int* rowmap,
                        hipStream_t s) {
    const size_t total = (size_t)Npad * (K >> 3);
    const int blocks = (int)((total + 255) / 256);
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((pack_weight_k<T>), dim3(blocks), dim3(256), 0, s, src, (T*)dst, N, K,
                                                Npad, rowmap));
}

// fp8 weights: one workgroup per output row. scale = absmax / 448 (e4m3 max), q = RNE(w * (448 / absmax)); stored in the
// 64-deep fragment order Wq[n_tile16][k_chunk64][lane][16]: lane (g<<4)|r <-> W[16*nt + r][64*kc + 16*g .. +16], so one
// wave-wide 16-byte load feeds TWO mfma_16x16x32 (bytes 0..7 and 8..15 of every lane; the activation fragment is read at the
// matching k offsets). With dst != null also writes the dequantised model-dtype copy T(q * scale) in the standard order (kernel test
// hooks; the engine keeps ONLY the fp8 bytes: prefill and batch >= 3 decode multiply fp8 x fp8, batch <= 2 expands in registers).
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_fp8_k(const float* __restrict__ src, unsigned char* __restrict__ dst8,
                                                         float* __restrict__ scale, T* __restrict__ dst, int N, int K, int Npad) {
    __shared__ float red[32];
    const int i = blockIdx.x;                              // output row (padded rows are zero)
    const bool real = i < N;
    float mx = 0.f;
    if (real) for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, fabsf(src[(size_t)i * K + k]));
    mx = block_max(mx, red);
    const float sc = mx > 0.f ? mx / 448.0f : 1.0f, inv = mx > 0.f ? 448.0f / mx : 1.0f;
    if (threadIdx.x == 0) scale[i] = sc;
    const int KC8 = K >> 6, KC = K >> 5, nt = i >> 4, r = i & 15;
    for (int q = threadIdx.x; q < (K >> 4); q += blockDim.x) {          // 16 consecutive k per step
        float w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = real ? src[(size_t)i * K + (size_t)q * 16 + j] * inv : 0.f;
        unsigned d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int v = 0;
            v = __builtin_amdgcn_cvt_pk_fp8_f32(w[4 * j], w[4 * j + 1], v, false);
            v = __builtin_amdgcn_cvt_pk_fp8_f32(w[4 * j + 2], w[4 * j + 3], v, true);
            d[j] = (unsigned)v;
        }
        const int kc = q >> 2, g = q & 3;
        reinterpret_cast<u4*>(dst8)[((size_t)nt * KC8 + kc) * 64 + (g * 16 + r)] = (u4){d[0], d[1], d[2], d[3]};
        // dequantised copy (test hooks only; dst == null in the engine), standard order: chunks of 8 k -> ((nt*KC + k/32)*64 + ((k%32)/8)*16 + r)
        if (dst)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            typename Vec8<T>::type v;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)d[2 * h + j], false);
                const auto f1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)d[2 * h + j], true);
                v[4 * j] = fromf<T>(f0[0] * sc); v[4 * j + 1] = fromf<T>(f0[1] * sc);
                v[4 * j + 2] = fromf<T>(f1[0] * sc); v[4 * j + 3] = fromf<T>(f1[1] * sc);
            }
            const int k0 = q * 16 + h * 8;
            reinterpret_cast<u4*>(dst)[((size_t)nt * KC + (k0 >> 5)) * 64 + (((k0 & 31) >> 3) * 16 + r)] = as_u4<T>(v);
        }
    }
}

void launch_pack_weight_fp8(int dtype, const float* src, void* dst8, float* scale, void* dst, int N, int K, int Npad, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((pack_weight_fp8_k<T>), dim3(Npad), dim3(256), 0, s, src, (unsigned char*)dst8,
                                                scale, (T*)dst, N, K, Npad));
}

// ------------------------------------------------------------------------------------------------------------------
// skinny GEMM (body in skinny_body.h)
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int MT, int EPI, bool NORM, int WAVES, bool XLDS, bool W8 = false>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 8 : 1) void skinny_gemm_k(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    skinny_tile<T, MT, EPI, NORM, WAVES, XLDS, NoWait, false, W8>(a, blockIdx.x, gridDim.x, dyn_smem, NoWait());
}

template <typename T, int MT, bool NORM, int WAVES, bool XLDS, bool W8 = false>
static void launch_skinny_epi(const GemmArgs& a, int epi, hipStream_t s) {
    const int nt = (a.N + 15) / 16;
    dim3 grid(nt), block(WAVES * 64);
    const size_t dyn = XLDS ? (size_t)a.M * a.K * 2 : 0;
#define RDX_SK(E) hipLaunchKernelGGL((skinny_gemm_k<T, MT, E, NORM, WAVES, XLDS, W8>), grid, block, dyn, s, a)
    switch (epi) {
        case EPI_NONE: RDX_SK(EPI_NONE); break;
        case EPI_RELU: RDX_SK(EPI_RELU); break;
        case EPI_GELU: RDX_SK(EPI_GELU); break;
        case EPI_RESID: RDX_SK(EPI_RESID); break;
        case EPI_SILU_MUL: RDX_SK(EPI_SILU_MUL); break;
        case EPI_LOGITS: RDX_SK(EPI_LOGITS); break;
        default: break;
    }
#undef RDX_SK
}

// activations fit the LDS staging path when M*K*2 bytes <= 32 KiB (keeps >= 2 workgroups per CU resident)
bool skinny_fits_lds(int M, int K) { return M <= 16 && (size_t)M * K * 2 <= 32 * 1024; }

template <typename T>
static void launch_skinny_T(const GemmArgs& a, int epi, hipStream_t s) {
    const bool norm = a.norm_w != nullptr;
    constexpr int WV = 8;
    // few output tiles (N <= 4096 -> at most one workgroup per CU): 16 waves per workgroup put twice as many
    // weight loads in flight per CU
    constexpr bool wide = true;
    const bool w8 = a.W8 && a.wscale && skinny_fits_lds(a.M, a.K) && a.K % 64 == 0;   // fp8 weight stream (else: W, which only the test hooks provide)
    if (wide && skinny_fits_lds(a.M, a.K) && (a.N + 15) / 16 <= 256 && a.K >= 4096) {
        if (w8) { if (norm) launch_skinny_epi<T, 1, true, 16, true, true>(a, epi, s); else launch_skinny_epi<T, 1, false, 16, true, true>(a, epi, s); return; }
        if (norm) launch_skinny_epi<T, 1, true, 16, true>(a, epi, s); else launch_skinny_epi<T, 1, false, 16, true>(a, epi, s);
        return;
    }
    // many output tiles: 4-wave workgroups at <= 64 VGPRs keep up to 2048 tiles resident at once (8 per CU), so the
    // whole GEMV runs as ONE round of workgroups sharing HBM evenly instead of 2-4 quantised rounds
    constexpr bool smallwg = true;
    if (smallwg && skinny_fits_lds(a.M, a.K) && (a.N + 15) / 16 > 512 && (size_t)a.M * a.K * 2 <= 16 * 1024) {
        if (w8) { if (norm) launch_skinny_epi<T, 1, true, 4, true, true>(a, epi, s); else launch_skinny_epi<T, 1, false, 4, true, true>(a, epi, s); return; }
        if (norm) launch_skinny_epi<T, 1, true, 4, true>(a, epi, s); else launch_skinny_epi<T, 1, false, 4, true>(a, epi, s);
        return;
    }
    if (skinny_fits_lds(a.M, a.K)) {
        if (w8) { if (norm) launch_skinny_epi<T, 1, true, WV, true, true>(a, epi, s); else launch_skinny_epi<T, 1, false, WV, true, true>(a, epi, s); return; }
        if (norm) launch_skinny_epi<T, 1, true, WV, true>(a, epi, s); else launch_skinny_epi<T, 1, false, WV, true>(a, epi, s);
    } else if (a.M <= 16) {
        launch_skinny_epi<T, 1, false, WV, false>(a, epi, s);     // caller pre-normalises (rmsnorm_k) when needed
    } else {
        launch_skinny_epi<T, 2, false, WV, false>(a, epi, s);
    }
}

void launch_skinny_gemm(int dtype, const GemmArgs& a, int epi, hipStream_t s) {
    // (fp8 weights: only with the e4m3 activation block, xpacked 4 -- the batch >= 3 rule of the fp8 scheme; callers go through skinny())
    if (xstat32_supported(a, epi) && (!(a.W8 && a.wscale) || (a.xpacked == 4 && a.xscale))) { launch_xstat32(dtype, a, epi, s); return; }
    RDX_DISPATCH_T(dtype, T, launch_skinny_T<T>(a, epi, s));
}

// ------------------------------------------------------------------------------------------------------------------
// tiled GEMM (plain or implicit-GEMM conv gather)
// ------------------------------------------------------------------------------------------------------------------
constexpr int TG_BM = 128, TG_BN = 128;

template <typename T>
__device__ __forceinline__ u4 gather_a(const T* X, const GemmArgs& a, const ConvGeom& cg, bool rowok, size_t rowbase,
                                       int ih0, int iw0, int k) {
    if (!rowok) return (u4){0u, 0u, 0u, 0u};
    if (cg.mode == 0) return ldg16(X + rowbase + k);
    const int kpos = k / cg.Cin, c0 = k - kpos * cg.Cin;
    const int kh = kpos / cg.KW, kw = kpos - kh * cg.KW;
    const int ih = ih0 + kh, iw = iw0 + kw;
    if (ih < 0 || ih >= cg.Hin || iw < 0 || iw >= cg.Win) return (u4){0u, 0u, 0u, 0u};
    return ldg16(X + rowbase + ((size_t)ih * cg.Win + iw) * cg.Cin + c0);
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void tiled_gemm_k(GemmArgs a, ConvGeom cg) {
    typedef typename Vec8<T>::type V8;
    __shared__ __attribute__((aligned(16))) u4 Xs[2][8][64];      // [buf][m_tile][lane] fragment order, 16 KiB

    // XCD-aware tile order: ids that land on one XCD (id % 8) walk consecutive m-blocks of one n-block, so the
    // weight panel is re-read from that XCD's own L2.
    const int MB = (a.M + TG_BM - 1) / TG_BM, NB = (a.N + TG_BN - 1) / TG_BN;
    const int nwg = MB * NB;
    int tile;
    {
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int bn = tile / MB, bm = tile - bn * MB;
    const int M0 = bm * TG_BM, N0 = bn * TG_BN;

    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int KC = a.K >> 5;
    const int NT16 = (a.N + 15) >> 4;
    const T* X = reinterpret_cast<const T*>(a.X);

    // staging rows of this thread: pass p -> m_tile = p*4 + w, row = M0 + m_tile*16 + r, k segment g
    bool s_ok[2];
    size_t s_base[2];
    int s_ih0[2], s_iw0[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = M0 + (p * 4 + w) * 16 + r;
        s_ok[p] = row < a.M;
        s_base[p] = 0; s_ih0[p] = 0; s_iw0[p] = 0;
        if (s_ok[p]) {
            if (cg.mode == 0) {
                s_base[p] = (size_t)row * a.ldx;
            } else {
                const int hw = cg.Hout * cg.Wout;
                const int b = row / hw, rem = row - b * hw;
                const int oh = rem / cg.Wout, ow = rem - oh * cg.Wout;
                s_base[p] = (size_t)b * cg.Hin * cg.Win * cg.Cin;
                s_ih0[p] = oh * cg.stride - cg.pad;
                s_iw0[p] = ow * cg.stride - cg.pad;
            }
        }
    }
    // weight fragment pointers (4 n-tiles of this wave)
    const u4* wptr[4];
    bool w_ok[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int tile16 = (N0 >> 4) + wn * 4 + nt;
        w_ok[nt] = tile16 < NT16;
        wptr[nt] = reinterpret_cast<const u4*>(a.W) + ((size_t)(w_ok[nt] ? tile16 : 0) * KC) * 64 + lane;
    }

    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    u4 xa[2], wf[4], wf_n[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) xa[p] = gather_a<T>(X, a, cg, s_ok[p], s_base[p], s_ih0[p], s_iw0[p], g * 8);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = w_ok[nt] ? ldg16(wptr[nt]) : (u4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 2; ++p) Xs[0][p * 4 + w][lane] = xa[p];
    __syncthreads();

    for (int c = 0; c < KC; ++c) {
        const int buf = c & 1;
        const bool more = (c + 1) < KC;
        if (more) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
                xa[p] = gather_a<T>(X, a, cg, s_ok[p], s_base[p], s_ih0[p], s_iw0[p], (c + 1) * 32 + g * 8);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf_n[nt] = w_ok[nt] ? ldg16(wptr[nt] + (size_t)(c + 1) * 64) : (u4){0u, 0u, 0u, 0u};
        }
        V8 xf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xf[mt] = as_vec8<T>(Xs[buf][wm * 4 + mt][lane]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma16(as_vec8<T>(wf[nt]), xf[mt], acc[nt][mt]);
        if (more) {
#pragma unroll
            for (int p = 0; p < 2; ++p) Xs[buf ^ 1][p * 4 + w][lane] = xa[p];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = wf_n[nt];
        }
        __syncthreads();
    }

    // epilogue: lane (r = m_local, g) holds out[m][n0 + g*4 + 0..3]
    T* out = reinterpret_cast<T*>(a.out);
    const T* resid = reinterpret_cast<const T*>(a.resid);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = N0 + (wn * 4 + nt) * 16 + g * 4;
        if (n >= a.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
            bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = M0 + (wm * 4 + mt) * 16 + r;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][e] + bv[e];
            if (EPI == EPI_SILU_MUL) {
                // gate rows live at n_local 0..7 (g = 0,1), up rows at 8..15 (g = 2,3): partner lane = lane ^ 32
                float u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = __shfl_xor(v[e], 32, 64);
                if (g < 2 && m < a.M) {
                    typedef T T4 __attribute__((ext_vector_type(4)));
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fromf<T>(swiglu<T>(v[e], u[e]));
                    const int oc = (N0 >> 1) + (wn * 4 + nt) * 8 + g * 4;
                    *reinterpret_cast<T4*>(out + (size_t)m * a.ldo + oc) = o;
                }
                continue;
            }
            if (m >= a.M) continue;
            if (EPI == EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if (EPI == EPI_RESID || EPI == EPI_RESID_RELU) {
                typedef T T4 __attribute__((ext_vector_type(4)));
                const T4 rv = *reinterpret_cast<const T4*>(resid + (size_t)m * a.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = tof<T>(rv[e]) + rnd<T>(v[e]);
                    if (EPI == EPI_RESID_RELU) v[e] = fmaxf(v[e], 0.f);
                }
            }
            typedef T T4 __attribute__((ext_vector_type(4)));
            T4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fromf<T>(v[e]);
            *reinterpret_cast<T4*>(out + (size_t)m * a.ldo + n) = o;
        }
    }
}

template <typename T>
static void launch_tiled_T(const GemmArgs& a, const ConvGeom& cg, int epi, hipStream_t s) {
    const int MB = (a.M + TG_BM - 1) / TG_BM, NB = (a.N + TG_BN - 1) / TG_BN;
    dim3 grid(MB * NB), block(256);
#define RDX_TG(E) hipLaunchKernelGGL((tiled_gemm_k<T, E>), grid, block, 0, s, a, cg)
    switch (epi) {
        case EPI_NONE: RDX_TG(EPI_NONE); break;
        case EPI_RELU: RDX_TG(EPI_RELU); break;
        case EPI_GELU: RDX_TG(EPI_GELU); break;
        case EPI_RESID: RDX_TG(EPI_RESID); break;
        case EPI_RESID_RELU: RDX_TG(EPI_RESID_RELU); break;
        case EPI_SILU_MUL: RDX_TG(EPI_SILU_MUL); break;
        default: break;
    }
#undef RDX_TG
}

void launch_tiled_gemm(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, launch_tiled_T<T>(a, cg, epi, s));
}

}  // namespace rdx

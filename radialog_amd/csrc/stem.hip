// Fused ResNet stem: 7x7 stride-2 convolution (+ folded BatchNorm bias, ReLU) and the 3x3 stride-2 pad-1 max pool in ONE launch
// (torchvision ResNet: conv1 -> bn1 -> relu -> maxpool, behind biovil_t/resnet.py:25-47).
//
// Separately the stem wrote its 224 x 224 x 64 output (6.4 MB per image) and the pool read it back: 270 us of a 5 ms batch-32 encode for
// 30 GFLOP. Here a workgroup owns a tile of PH x PW = 4 x 16 POOLED pixels, computes the (2 PH + 1) x (2 PW + 1) = 9 x 33 conv pixels
// it depends on (halo recomputed: 1.7x the MFMA work, which is nothing), keeps them in LDS, pools and writes 64 x stem channels once.
//   * input: the zero-padded NHWC4 image of img_prep_k ([B][S + 6][S + 6][4], 4th channel zero); K = 7 (kh) x 8 (kw, last zero) x 4 (c)
//     = 224 = 7 MFMA k-chunks, one per kh; lane (g, r) of a B fragment = conv pixel r of a 16-pixel row segment, taps kw = 2g, 2g + 1
//     = 16 contiguous bytes, and consecutive pixels are 16 bytes apart: a quarter wave reads 256 contiguous bytes;
//   * weights (NT x 7 fragments) live in registers for the whole kernel; a wave walks conv-row segments (m-tiles of 16 pixels),
//     next segment's loads in flight under this one's MFMAs;
//   * conv pixels outside the image (the pool's padding) are stored as 0: after ReLU every value is >= 0 and every pool window holds
//     at least one real pixel, so max-with-0 equals torch's -inf padding;
//   * rounding points: T(relu(acc + bias)) for the conv output (as the two-kernel path), the pool is exact on those values.
#include <algorithm>

#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

constexpr int ST_PH = 4, ST_PW = 16, ST_CR = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1, ST_SEG = 3;      // conv rows, cols, 16-pixel segments per row

template <typename T, int NT>
__global__ __launch_bounds__(256) void stem_pool_k(const T* __restrict__ in, const u4* __restrict__ Wp, const float* __restrict__ bias,
                                                   T* __restrict__ out, int Hp, int Hc, int Ho, int packed_mt) {
    typedef typename Vec8<T>::type V8;
    typedef T T4 __attribute__((ext_vector_type(4)));
    constexpr int C = NT * 16, KC = 7, CPITCH = C + 8;             // conv tile pixel pitch in LDS (elements): + 16 B against bank conflicts
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);                          // [ST_CR][ST_SEG * 16][CPITCH]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, ph0 = blockIdx.y * ST_PH, pw0 = blockIdx.x * ST_PW;
    const int cy0 = 2 * ph0 - 1, cx0 = 2 * pw0 - 1;                // first conv row / column of the tile (may be -1)
    // weights -> registers (fragment order: [nt][kc][lane])
    u4 wf[NT][KC];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) wf[nt][kc] = ldg16(Wp + (size_t)(nt * KC + kc) * 64 + lane);
    float b4[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + nt * 16 + g * 4);
        b4[nt][0] = bv.x; b4[nt][1] = bv.y; b4[nt][2] = bv.z; b4[nt][3] = bv.w;
    }
    const T* img = in + (size_t)b * Hp * Hp * 4;
    // m-tile t = (conv row cr, segment sg): conv pixel (cy0 + cr, cx0 + 16 sg + r); taps start at padded (2 y, 2 x) (pad 3 folded in)
    auto load_seg = [&](int t, u4 (&xf)[KC]) {
        const int cr = t / ST_SEG, sg = t - cr * ST_SEG;
        const int cy = min(max(cy0 + cr, 0), Hc - 1), cx = min(max(cx0 + sg * 16 + r, 0), Hc - 1);      // clamped: out-of-image pixels are zeroed at the store
        const T* p = img + ((size_t)(2 * cy) * Hp + 2 * cx + 2 * g) * 4;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) xf[kc] = ldg16(p + (size_t)kc * Hp * 4);
    };
    constexpr int NTILES = ST_CR * ST_SEG;
    u4 xa[KC], xb[KC];
    int t = w;
    if (t < NTILES) load_seg(t, xa);
    for (; t < NTILES; t += 4) {
        if (t + 4 < NTILES) load_seg(t + 4, xb);
        v4f acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(as_vec8<T>(wf[nt][kc]), as_vec8<T>(xa[kc]), acc[nt]);
        const int cr = t / ST_SEG, sg = t - cr * ST_SEG;
        const int cy = cy0 + cr, cx = cx0 + sg * 16 + r;
        const bool inside = cy >= 0 && cy < Hc && cx >= 0 && cx < Hc;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            T4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fromf<T>(inside ? fmaxf(acc[nt][e] + b4[nt][e], 0.f) : 0.f);
            *reinterpret_cast<T4*>(tile + ((size_t)cr * (ST_SEG * 16) + sg * 16 + r) * CPITCH + nt * 16 + g * 4) = o;
        }
        if (t + 4 < NTILES) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) xa[kc] = xb[kc];
        }
    }
    __syncthreads();
    // pool: work item = (pooled pixel, 8-channel group); window rows 2 ph .. 2 ph + 2, cols 2 pw .. 2 pw + 2 of the conv tile
    constexpr int C8 = C / 8;
    for (int i = threadIdx.x; i < ST_PH * ST_PW * C8; i += 256) {
        const int c8 = i % C8, pp = i / C8, pw = pp % ST_PW, ph = pp / ST_PW;
        if (ph0 + ph >= Ho || pw0 + pw >= Ho) continue;
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const V8 v = as_vec8<T>(*reinterpret_cast<const u4*>(tile + ((size_t)(2 * ph + dy) * (ST_SEG * 16) + 2 * pw + dx) * CPITCH + c8 * 8));
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], tof<T>(v[j]));
            }
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fromf<T>(m[j]);
        const size_t pix = ((size_t)b * Ho + ph0 + ph) * Ho + pw0 + pw;
        if (packed_mt)      // fragment-packed output [C / 32][packed_mt][lane (g = c8 % 4, r = m % 16)][8] for pconv_k (round 4)
            stg16(reinterpret_cast<u4*>(out) + ((size_t)(c8 >> 2) * packed_mt + (pix >> 4)) * 64 + (c8 & 3) * 16 + (pix & 15), as_u4<T>(o));
        else
            stg16(out + pix * C + c8 * 8, as_u4<T>(o));
    }
}

bool stem_pool_supported(int stem_channels) {
    return stem_channels == 64 || stem_channels == 32;
}

// in: padded NHWC4 image [B][Hp][Hp][4] (Hp = S + 6); Wp: the stem weights packed as a [stem][224] GEMM weight; out [B][Ho][Ho][stem]
void launch_stem_pool(int dtype, const void* in, const void* Wp, const float* bias, void* out, int B, int Hp, int Hc, int Ho, int stem,
                      int packed_mt, hipStream_t s) {
    dim3 grid((Ho + ST_PW - 1) / ST_PW, (Ho + ST_PH - 1) / ST_PH, B), block(256);
    RDX_DISPATCH_T(dtype, T, {
        if (stem == 64) {
            const size_t smem = (size_t)ST_CR * ST_SEG * 16 * (64 + 8) * sizeof(T);
            static DevOnce attr;
            if (attr.first()) { (void)hipFuncSetAttribute((const void*)stem_pool_k<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); }
            hipLaunchKernelGGL((stem_pool_k<T, 4>), grid, block, smem, s, (const T*)in, (const u4*)Wp, bias, (T*)out, Hp, Hc, Ho, packed_mt);
        } else {
            const size_t smem = (size_t)ST_CR * ST_SEG * 16 * (32 + 8) * sizeof(T);
            hipLaunchKernelGGL((stem_pool_k<T, 2>), grid, block, smem, s, (const T*)in, (const u4*)Wp, bias, (T*)out, Hp, Hc, Ho, packed_mt);
        }
    });
}

}  // namespace rdx

// ---- microbenchmark: how many bytes per second does ONE CU pull from L2 / MALL? ("what bounds the tile GEMMs") ----------
// every workgroup reads `bytes_per_wg` (its own region, or all the same one) `reps` times with 16-byte lane loads, 8 in flight per
// lane; mode 0 = global_load_dwordx4 into registers, mode 1 = global_load_lds_dwordx4 (LDS-DMA) into a 64 KiB LDS ring.
namespace rdx {
typedef __attribute__((address_space(1))) const void* lb_gptr_t;
typedef __attribute__((address_space(3))) void* lb_lptr_t;
template <int MODE>
__global__ __launch_bounds__(256) void l2_bench_k(const u4* __restrict__ buf, size_t u4_per_wg, int shared, int reps, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) u4 lbuf[];
    const u4* base = buf + (shared ? 0 : (size_t)blockIdx.x * u4_per_wg);
    unsigned acc = 0;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int rep = 0; rep < reps; ++rep) {
        for (size_t i = threadIdx.x; i + 7 * 256 < u4_per_wg; i += 8 * 256) {
            if (MODE == 0) {
                u4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ldg16(base + i + j * 256);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_global_load_lds((lb_gptr_t)(base + i + j * 256), (lb_lptr_t)(lbuf + (j * 4 + w) * 64), 16, 0, 0);
            }
        }
        if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc ^= lbuf[threadIdx.x].x; }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
void launch_l2_bench(int mode, const void* buf, size_t bytes_per_wg, int shared, int reps, int wgs, unsigned* sink, hipStream_t s) {
    if (mode == 0) hipLaunchKernelGGL((l2_bench_k<0>), dim3(wgs), dim3(256), 0, s, (const u4*)buf, bytes_per_wg / 16, shared, reps, sink);
    else hipLaunchKernelGGL((l2_bench_k<1>), dim3(wgs), dim3(256), 32 * 1024, s, (const u4*)buf, bytes_per_wg / 16, shared, reps, sink);
}
}  // namespace rdx

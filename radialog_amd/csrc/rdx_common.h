// Shared device/host helpers for librdx (gfx950 / CDNA4 only: 64-wide wavefronts, MFMA 16x16x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace rdx {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

enum DType { DT_F16 = 0, DT_BF16 = 1 };

// ---- scalar conversions (round-to-nearest-even, the rounding torch's .to(half/bfloat16) applies) --------------
template <typename T> __device__ __forceinline__ float tof(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T fromf(float x) { return (T)x; }
// round a float through T and back: the "every torch op rounds its output to the model dtype" emulation
template <typename T> __device__ __forceinline__ float rnd(float x) { return (float)((T)x); }

template <typename T> struct Vec8;           // 8 x T in one 16-byte register quad
template <> struct Vec8<f16> { typedef v8h type; };
template <> struct Vec8<bf16> { typedef v8b type; };

template <typename T> __device__ __forceinline__ typename Vec8<T>::type as_vec8(u4 v) {
    return __builtin_bit_cast(typename Vec8<T>::type, v);
}
template <typename T> __device__ __forceinline__ unsigned short bits16(T v) { return __builtin_bit_cast(unsigned short, v); }
template <typename T> __device__ __forceinline__ T from_bits16(unsigned short b) { return __builtin_bit_cast(T, b); }
template <typename T> __device__ __forceinline__ u4 as_u4(typename Vec8<T>::type v) { return __builtin_bit_cast(u4, v); }

__device__ __forceinline__ v4f mfma16(v8h a, v8h b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f mfma16(v8b a, v8b b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// 16-byte GLOBAL-memory loads/stores. `ldg_nt` = streamed-once data (decode weights): non-temporal policy.
// The explicit address_space(1) cast matters: a pointer fetched from a device-memory table (chain.hip) is generic, and
// generic accesses become FLAT instructions, whose completion order is not guaranteed -- the compiler then falls back
// to s_waitcnt vmcnt(0) everywhere and a software-pipelined weight stream collapses to one batch in flight.
typedef __attribute__((address_space(1))) u4 g_u4;
__device__ __forceinline__ u4 ldg16(const void* p) { return *(const g_u4*)p; }
__device__ __forceinline__ u4 ldg16_nt(const void* p) { return __builtin_nontemporal_load((const g_u4*)p); }
__device__ __forceinline__ void stg16(void* p, u4 v) { *(g_u4*)p = v; }

// ---- fp8 (OCP e4m3) -> model dtype, exact (every e4m3 value is representable in bf16 and f16) --------------------------------
// two dwords = 8 fp8 bytes -> 8 model-dtype values in one 16-byte register quad (an MFMA A-operand chunk)
template <typename T> __device__ __forceinline__ u4 dequant8(unsigned lo, unsigned hi);
template <> __device__ __forceinline__ u4 dequant8<bf16>(unsigned lo, unsigned hi) {
    const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const auto c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    // bf16 = upper half of the f32 (exact here): (f0.hi16) | (f1.hi16 << 16). The elements are copied to scalars first:
    // __builtin_bit_cast on a vector ELEMENT (a[1]) reads element 0 with this compiler.
    const float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1], c0 = c[0], c1 = c[1], d0 = d[0], d1 = d[1];
    return (u4){__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, a0), 0x07060302u),
                __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b1), __builtin_bit_cast(unsigned, b0), 0x07060302u),
                __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, c1), __builtin_bit_cast(unsigned, c0), 0x07060302u),
                __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, d1), __builtin_bit_cast(unsigned, d0), 0x07060302u)};
}
template <> __device__ __forceinline__ u4 dequant8<f16>(unsigned lo, unsigned hi) {
    const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const auto c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    return (u4){__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a[0], a[1])), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b[0], b[1])),
                __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(c[0], c[1])), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[0], d[1]))};
}

// ---- model dtype -> fp8 (OCP e4m3) activation quantisation of the fp8 path: q = RNE_e4m3(x * (448 / absmax)), scale = absmax / 448 ----------
// (the arithmetic of pack_weight_fp8_k and of the oracle's fake quantisation; an all-zero range gets scale 1)
__device__ __forceinline__ void fp8_scale(float amax, float& sc, float& inv) {
    sc = amax > 0.f ? amax / 448.0f : 1.0f;
    inv = amax > 0.f ? 448.0f / amax : 1.0f;
}
template <typename T> __device__ __forceinline__ u2 quant8(const typename Vec8<T>::type& v, float inv) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(tof<T>(v[0]) * inv, tof<T>(v[1]) * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(tof<T>(v[2]) * inv, tof<T>(v[3]) * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(tof<T>(v[4]) * inv, tof<T>(v[5]) * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(tof<T>(v[6]) * inv, tof<T>(v[7]) * inv, hi, true);
    return (u2){(unsigned)lo, (unsigned)hi};
}
template <typename T> __device__ __forceinline__ float amax8(const typename Vec8<T>::type& v) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(tof<T>(v[j])));
    return m;
}
// ---- wave (64 lanes) reductions ----------------------------------------------------------------------------------
// Cross-lane traffic goes through DPP / v_readlane / v_permlane*_swap (VALU speed, ~8 cycles each) rather than
// __shfl_xor (ds_bpermute: an LDS-crossbar round trip of ~100 cycles per step) -- these reductions sit on the critical
// path of latency-bound kernels (decode attention, the RMSNorm prologue of every GEMV).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
// sum / max over each aligned group of 16 lanes (one DPP row); every lane of the group gets the result
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane_const) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_const));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] (gfx950 v_permlane16_swap / v_permlane32_swap: odd rows of the first operand
// are exchanged with even rows of the second, so with both operands = v the pair holds {even rows, odd rows} twice)
__device__ __forceinline__ float xor16_sum(float v) {
    const int b = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const int b = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}

// block reductions through a small LDS scratch (>= 32 floats); all threads get the result.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// K-cache layout inside one (row, head) slab [max_len][128]: every group of 16 positions is stored in the MFMA A-fragment
// order [dim / 32][lane (g = (dim % 32) / 8, r = pos % 16)][8], so that a wave's fragment load for the scores (16 positions x
// 32 dims) is ONE contiguous KiB instead of 16 rows x 16 bytes per quarter wave (64 tag look-ups per load instruction).
// `dim` is a multiple of 8; max_len is a multiple of 16. V stays row-major (its reads are row-contiguous).
// The order is chosen per context (LlamaDims::k_perm): contexts of up to 8 rows (the 16-wave latency attention: batch-1 attention
// 8.7 -> 7.3 us) use it, larger ones keep K row-major (the 4-wave throughput attention measured 1.5 us per launch slower with it).
__host__ __device__ __forceinline__ size_t kperm(int pos, int dim, int perm = 1) {
    return perm ? (size_t)(pos >> 4) * 2048 + (size_t)((((dim >> 5) * 64) + ((dim & 31) >> 3) * 16 + (pos & 15)) << 3)
                : (size_t)pos * 128 + dim;
}

// packed GEMM weight geometry: [n_tile16][k_chunk32][64 lanes][8 elems]; lane = (g<<4)|r holds
// W[n_tile*16 + r][k_chunk*32 + g*8 .. +8]  -- the MFMA 16x16x32 operand fragment, 1 KiB per block.
__host__ __device__ __forceinline__ size_t packed_elems(int n, int k) { return (size_t)((n + 15) / 16) * 16 * (size_t)k; }

}  // namespace rdx

// ---- host side ---------------------------------------------------------------------------------------------------
// hipFuncSetAttribute is per DEVICE: a process may own contexts on several GPUs (one per GPU is the rule, but nothing enforces it), so the
// "set once" guards of the launchers are keyed by the current device
namespace rdx {
struct DevOnce {
    bool done[64] = {};
    bool first() { int d = 0; (void)hipGetDevice(&d); d &= 63; if (done[d]) return false; done[d] = true; return true; }
};
}  // namespace rdx

#define RDX_DISPATCH_T(dt, T, ...)                         \
    do {                                                   \
        if ((dt) == rdx::DT_F16) { typedef rdx::f16 T; __VA_ARGS__; } \
        else { typedef rdx::bf16 T; __VA_ARGS__; }         \
    } while (0)

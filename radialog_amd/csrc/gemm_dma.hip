// Large-M MFMA GEMM with LDS-DMA staging (gfx950):  out[M,N] = epilogue(X[M,K] . W[N,K]^T),  K % 64 == 0.
//
// Used for the Llama prefill GEMMs, the 1x1 convolutions of the ResNet trunk / projector and the Q-Former at batch > 1
// (everything `tiled_gemm_k` does except the im2col-gather convolutions). Structure:
//   * 128 x 128 block tile, 4 waves as 2 (M) x 2 (N), each wave 4 x 4 MFMA 16x16x32 tiles, BK = 64 per step;
//   * both operands go global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip), 1 KiB per wave-instruction.
//     Weights are already stored in MFMA-fragment order, so a packed block lands in LDS exactly as ds_read_b128 wants
//     it (lane-linear, conflict-free). For the activations the per-lane SOURCE address follows the fragment order
//     (lane (r,g) fetches row r, k-octet g of a 16 x 32 sub-tile), so their LDS image is fragment-linear as well;
//   * two LDS buffers (2 x 32 KiB); the loads of step s+1 are in flight across the barrier while step s is multiplied:
//     counted `s_waitcnt vmcnt(8)` + raw `s_barrier` (a __syncthreads() would drain the DMA queue);
//   * split-K over blockIdx.z for small grids (prefill at M = 160 has only 64-350 output tiles for 256 CUs): fp32
//     partial slabs + a fixed-order reduce kernel that applies the epilogue (deterministic, no atomics).
#include <algorithm>
#include <stdlib.h>

#include "rdx_common.h"
#include "rdx_kernels.h"
#include "skinny_body.h"

namespace rdx {

constexpr int DG_BM = 128, DG_BN = 128, DG_BK = 64;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// epilogue on 4 consecutive columns n..n+3 of row m (shared by the main kernel and the split-K reducer)
template <typename T, int EPI>
__device__ __forceinline__ void store4(const GemmArgs& a, int m, int n, float v[4]) {
    typedef T T4 __attribute__((ext_vector_type(4)));
    T* out = reinterpret_cast<T*>(a.out);
    if (a.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    if (EPI == EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    } else if (EPI == EPI_RESID || EPI == EPI_RESID_RELU) {
        const T4 rv = *reinterpret_cast<const T4*>(reinterpret_cast<const T*>(a.resid) + (size_t)m * a.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = tof<T>(rv[e]) + rnd<T>(v[e]);
            if (EPI == EPI_RESID_RELU) v[e] = fmaxf(v[e], 0.f);
        }
    }
    T4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fromf<T>(v[e]);
    *reinterpret_cast<T4*>(out + (size_t)m * a.ldo + n) = o;
}

// CONV: the activation operand is an implicit-GEMM gather from an NHWC tensor (K ordered (kh, kw, c), Cin % 8 == 0, so a lane's
// 16-byte piece lies inside one filter tap): every lane hands global_load_lds its own source address; taps that fall
// into the zero padding read a 16-byte zero block instead.
// Two LDS stages of 32 KiB (64 KiB: two workgroups per CU; a four-stage ring for grids that do not fill the chip was measured in round 2
// and removed -- no gain).
template <typename T, int EPI, bool SPLIT, bool CONV = false>
__global__ __launch_bounds__(256) void gemm_dma_k(GemmArgs a, float* __restrict__ partial, int steps_per_split, ConvGeom cg = ConvGeom(),
                                                  const void* zero16 = nullptr) {
    typedef typename Vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) u4 lds[];          // [stage 2][operand 2][block 16][lane 64]
    const int MB = (a.M + DG_BM - 1) / DG_BM, NB = (a.N + DG_BN - 1) / DG_BN;
    const int nwg = MB * NB;
    int tile;
    {   // XCD-aware order: ids that land on one XCD walk consecutive m-blocks of one n-block (weight panel stays in its L2)
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    const int bn = tile / MB, bm = tile - bn * MB;
    const int M0 = bm * DG_BM, N0 = bn * DG_BN;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int KC = a.K >> 5, NT16 = (a.N + 15) >> 4;
    const int nsteps_total = a.K / DG_BK;
    const int s0 = blockIdx.z * steps_per_split;
    const int s1 = min(nsteps_total, s0 + steps_per_split);
    const T* X = reinterpret_cast<const T*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W);

    // this wave stages blocks i = w*4 .. w*4+3 of each operand per step; block i = (sub-tile i>>1, k-chunk i&1)
    const u4* wsrc[4];
    const T* xsrc[4];
    int ih0[4], iw0[4];                              // CONV: top-left input coordinate of this lane's output pixel
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = w * 4 + j, st = i >> 1, kc = i & 1;
        const int t16 = min((N0 >> 4) + st, NT16 - 1);
        wsrc[j] = Wp + ((size_t)t16 * KC + kc) * 64 + lane;
        const int row = min(M0 + st * 16 + r, a.M - 1);
        if (!CONV) {
            xsrc[j] = X + (size_t)row * a.ldx + kc * 32 + g * 8;
            ih0[j] = iw0[j] = 0;
        } else {
            const int hw = cg.Hout * cg.Wout;
            const int b = row / hw, rem = row - b * hw;
            const int oh = rem / cg.Wout, ow = rem - oh * cg.Wout;
            xsrc[j] = X + (size_t)b * cg.Hin * cg.Win * cg.Cin;
            ih0[j] = oh * cg.stride - cg.pad;
            iw0[j] = ow * cg.stride - cg.pad;
        }
    }
    auto stage = [&](int s, int buf) {
        u4* base = lds + (size_t)buf * 2 * 16 * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = w * 4 + j;
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (size_t)s * 2 * 64), (lptr_t)(base + i * 64), 16, 0, 0);
            const T* xp;
            if (!CONV) {
                xp = xsrc[j] + (size_t)s * DG_BK;
            } else {
                const int k = s * DG_BK + (i & 1) * 32 + g * 8;
                const int kpos = k / cg.Cin, c0 = k - kpos * cg.Cin;
                const int kh = kpos / cg.KW, kw = kpos - kh * cg.KW;
                const int ih = ih0[j] + kh, iw = iw0[j] + kw;
                const bool inb = ih >= 0 && ih < cg.Hin && iw >= 0 && iw < cg.Win;
                xp = inb ? xsrc[j] + ((size_t)ih * cg.Win + iw) * cg.Cin + c0 : reinterpret_cast<const T*>(zero16);
            }
            __builtin_amdgcn_global_load_lds((gptr_t)xp, (lptr_t)(base + (16 + i) * 64), 16, 0, 0);
        }
    };

    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    // debug timeline (rdx_kernel_bench with a trace buffer): [0] entry, [1] first stage landed, [2] k loop done, [3] end, [4] steps, [5] XCC
    long long* trc = (a.trace && threadIdx.x == 0) ? a.trace + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 : nullptr;
    if (trc) { trc[0] = (long long)__builtin_amdgcn_s_memrealtime(); trc[6] = (long long)__builtin_amdgcn_s_memtime(); }

    auto multiply = [&](const u4* base) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            V8 wf[4], xf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = as_vec8<T>(base[((wn * 4 + nt) * 2 + kc) * 64 + lane]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = as_vec8<T>(base[(16 + (wm * 4 + mt) * 2 + kc) * 64 + lane]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma16(wf[nt], xf[mt], acc[nt][mt]);
        }
    };
    {
        if (s0 < s1) stage(s0, 0);
        for (int s = s0; s < s1; ++s) {
            const int buf = (s - s0) & 1;
            if (s + 1 < s1) {
                stage(s + 1, buf ^ 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the 8 loads of step s have landed, step s+1 stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (trc && s == s0) trc[1] = (long long)__builtin_amdgcn_s_memrealtime();
            multiply(lds + (size_t)buf * 2 * 16 * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // everyone is done reading buf before it is re-staged
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (trc) { trc[2] = (long long)__builtin_amdgcn_s_memrealtime(); trc[4] = s1 - s0; trc[5] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }

    // epilogue: lane (r = m_local, g) holds out[m][n0 + g*4 + 0..3]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = N0 + (wn * 4 + nt) * 16 + g * 4;
        if (n >= a.N) continue;                                     // whole 16-column tile at once (N % 16 == 0)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = M0 + (wm * 4 + mt) * 16 + r;
            float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
            if (SPLIT) {
                if (m < a.M)
                    *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.z * a.M + m) * a.N + n) = make_float4(v[0], v[1], v[2], v[3]);
                continue;
            }
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
                    *reinterpret_cast<T4*>(reinterpret_cast<T*>(a.out) + (size_t)m * a.ldo + oc) = o;
                }
                continue;
            }
            if (m < a.M) store4<T, EPI>(a, m, n, v);
        }
    }
    if (trc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trc[3] = (long long)__builtin_amdgcn_s_memrealtime(); trc[7] = (long long)__builtin_amdgcn_s_memtime(); }
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 32 variant with a 4-stage LDS ring for large M (batched prefill, batched encoder GEMMs).
// PMC counters put gemm_dma_k at 33 % MFMA utilisation with the LDS 18 % busy and the waves resident: its single stage of
// DMA lookahead (one 128 x 128 x 64 step = ~0.2 us of MFMA work) is far shorter than the memory latency. Here a block
// tile of 256 x 256 halves the operand bytes per MFMA, a stage is 32 KiB (one k-chunk of 32 for 16 + 16 sub-tiles), and
// four stages are resident (128 KiB, one workgroup of 8 waves per CU): three stages = ~1.3 us of MFMA work are always
// in flight. Waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers, two waves per SIMD).
// One barrier per stage: it publishes stage s and at the same time frees the slot that stage s+3 is then loaded into.
// ------------------------------------------------------------------------------------------------------------------
// MTW = row tiles of 16 per wave: 8 -> 256-row block (32 KiB stages, 128 KiB ring); 10 -> 320-row block (36 KiB stages, 144 KiB ring, 160
// accumulator registers), for shapes where it saves a round: M = 5120, N = 4096 (o_proj / down of the batch-32 prefill) is 320 tiles of
// 256 x 256 -- two rounds on 256 CUs, the second a quarter full -- but exactly 256 tiles of 320 x 256.
constexpr int G2_BN = 256, G2_NS = 4;

template <typename T, int EPI, int MTW, int SP>
__global__ __launch_bounds__(512, 2) void gemm_dma256_k(GemmArgs a) {
    typedef typename Vec8<T>::type V8;
    constexpr int G2_BM = 32 * MTW, XS = 2 * MTW;                      // activation sub-tiles of 16 rows per stage
    constexpr int XPW = (XS + 7) / 8;                                   // ... staged per wave (the last waves re-stage sub-tile XS - 1: same bytes)
    constexpr int SUB = 16 + XS;                                        // KiB (sub-tiles) per stage
    extern __shared__ __attribute__((aligned(16))) u4 lds[];          // [stage 4][W 16 | X XS][lane 64]
    const int MB = (a.M + G2_BM - 1) / G2_BM, NB = (a.N + G2_BN - 1) / G2_BN;
    const int nwg = MB * NB;
    int tile;
    {   // XCD-aware order (see gemm_dma_k)
        const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
    }
    // the 32 workgroups an XCD runs at a time (consecutive `tile`s) form 4 row blocks x 8 column blocks: 12 operand panels per stage in its L2 instead of the
    // 22 of a column strip (fabric traffic of a 5120 x 22016 x 4096 launch 1.75 -> 0.89 GB in the fp8 twin, gemm8.hip). Bands of 8 column blocks, groups of 4
    // row blocks inside a band, column-major inside a group.
    const int band = tile / (MB * 8), idx = tile - band * MB * 8, wdt = min(8, NB - band * 8);
    const int gq = idx / (4 * wdt), gh = min(4, MB - gq * 4), rem = idx - gq * 4 * wdt;
    const int bn = band * 8 + rem / gh, bm = gq * 4 + rem % gh;
    const int M0 = bm * G2_BM, N0 = bn * G2_BN;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = w >> 2, wn = w & 3;
    const int KC = a.K >> 5, NT16 = (a.N + 15) >> 4;
    const int nsteps = KC;                                              // one k-chunk of 32 per stage
    const T* X = reinterpret_cast<const T*>(a.X);
    const u4* Wp = reinterpret_cast<const u4*>(a.W) + lane;

    // this wave stages weight sub-tiles 2w, 2w+1 and activation sub-tiles XPW w ..; addresses: one base per sub-tile, advanced by the stage index
    const u4* wsrc[2];
    const T* xsrc[XPW];
    int xst[XPW];
#pragma unroll
    for (int j = 0; j < 2; ++j) wsrc[j] = Wp + (size_t)min((N0 >> 4) + w * 2 + j, NT16 - 1) * KC * 64;
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        xst[j] = min(w * XPW + j, XS - 1);
        xsrc[j] = X + (size_t)min(M0 + xst[j] * 16 + r, a.M - 1) * a.ldx + g * 8;
    }
    constexpr int LPS = 2 + XPW;                                        // loads per wave per stage
    auto stage1 = [&](int s, int slot, int j) {                         // piece j of this wave's LPS pieces of stage s (j is a compile-time constant at every call)
        u4* base = lds + (size_t)slot * SUB * 64;
        if (j < 2) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (size_t)s * 64), (lptr_t)(base + (w * 2 + j) * 64), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[j - 2] + (size_t)s * 32), (lptr_t)(base + (16 + xst[j - 2]) * 64), 16, 0, 0);
    };
    auto stage = [&](int s, int slot) {                                 // s is clamped by the caller: loads are unconditional
#pragma unroll
        for (int j = 0; j < LPS; ++j) stage1(s, slot, j);
    };

    constexpr int spread = SP;                                          // placement of the next stage's LDS-DMA pieces (see the loop)
    v4f acc[4][MTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MTW; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int p = 0; p < G2_NS - 1; ++p) stage(min(p, nsteps - 1), p);   // stages 0..2 in flight
    for (int s = 0; s < nsteps; ++s) {
        if (LPS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // this wave's loads of stage s have landed (two younger stages may fly)
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // ... everyone's have, and everyone finished stage s-1
        // stage s + 3 goes into the slot stage s-1 was read from (past the end: the last stage again, harmless, keeps the wait counted). SP: its LDS-DMA
        // pieces are issued one by one behind the MFMAs of row tiles 1, 3, 5, ... instead of all in front of the first MFMA: an LDS-DMA instruction
        // costs 60-185 cycles of issue, and four or five of them up front left the matrix pipe idle that long every stage (batched prefill
        // 68.8 -> 66.5 ms; behind the even row tiles 66.2-67.4)
        const int sn = min(s + G2_NS - 1, nsteps - 1), slotn = (s + G2_NS - 1) % G2_NS;
        if (!spread) stage(sn, slotn);
        const u4* base = lds + (size_t)(s % G2_NS) * SUB * 64;
        V8 wf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = as_vec8<T>(base[(wn * 4 + nt) * 64 + lane]);
        V8 xf[MTW];
        xf[0] = as_vec8<T>(base[(16 + wm * MTW) * 64 + lane]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            if (mt + 1 < MTW) xf[mt + 1] = as_vec8<T>(base[(16 + wm * MTW + mt + 1) * 64 + lane]);   // the next row tile's fragment is on its way under these MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mfma16(wf[nt], xf[mt], acc[nt][mt]);
            if (spread && (mt & 1) == 1 && (mt >> 1) < LPS) stage1(sn, slotn, mt >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // LDS reads of this stage are done before the next barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // epilogue: lane (r = m_local, g) holds out[m][n0 + g*4 + 0..3]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = N0 + (wn * 4 + nt) * 16 + g * 4;
        if (n >= a.N) continue;                                     // whole 16-column tile at once (N % 16 == 0)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int m = M0 + (wm * MTW + mt) * 16 + r;
            float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
            if (EPI == EPI_SILU_MUL) {
                float u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = __shfl_xor(v[e], 32, 64);
                if (g < 2 && m < a.M) {
                    typedef T T4 __attribute__((ext_vector_type(4)));
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fromf<T>(swiglu<T>(v[e], u[e]));
                    const int oc = (N0 >> 1) + (wn * 4 + nt) * 8 + g * 4;
                    *reinterpret_cast<T4*>(reinterpret_cast<T*>(a.out) + (size_t)m * a.ldo + oc) = o;
                }
                continue;
            }
            if (m < a.M) store4<T, EPI>(a, m, n, v);
        }
    }
}

// split-K reducer: sums the fp32 slabs in split order and applies the epilogue. One thread = 4 consecutive columns.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_k(GemmArgs a, const float* __restrict__ partial, int splits) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N4 = a.N >> 2;
    if (idx >= (size_t)a.M * N4) return;
    const int m = (int)(idx / N4), n = (int)(idx % N4) * 4;
    auto sum4 = [&](int col, float v[4]) {
        v[0] = v[1] = v[2] = v[3] = 0.f;
        for (int s = 0; s < splits; ++s) {
            const float4 p = *reinterpret_cast<const float4*>(partial + ((size_t)s * a.M + m) * a.N + col);
            v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
        }
    };
    float v[4];
    if (EPI == EPI_SILU_MUL) {
        const int n_local = n & 15;
        if (n_local >= 8) return;                                    // gate half drives; up half is read at +8
        float u[4];
        sum4(n, v);
        sum4(n + 8, u);
        typedef T T4 __attribute__((ext_vector_type(4)));
        T4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fromf<T>(swiglu<T>(v[e], u[e]));
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(a.out) + (size_t)m * a.ldo + (n >> 4) * 8 + n_local) = o;
        return;
    }
    sum4(n, v);
    store4<T, EPI>(a, m, n, v);
}

template <typename T, int EPI>
static void launch_dma_epi(const GemmArgs& a_in, float* ws, size_t ws_floats, hipStream_t s) {
    const GemmArgs& a = a_in;
    {   // large M: the 256 x 256 tile kernel with the 4-stage ring, as long as its grid still covers the chip
        constexpr int big = 256;                                  // minimum number of 256 x 256 tiles
        const int MB2 = (a.M + 255) / 256, NB2 = (a.N + G2_BN - 1) / G2_BN;
        // (the short-K / narrow-N 1x1 convolutions of the encoder are memory-bound and do better with the small tile)
        if (big && a.M >= 1024 && a.K >= 512 && a.N >= 1024 && MB2 * NB2 >= big) {
            // 320-row blocks when they save rounds on the 256 CUs: cost = rounds x rows per block, with a 15 % handicap -- a 320-row tile is less
            // efficient than its size says (24 activation sub-tiles staged for 20, 200 VGPRs): the Q-Former cross-K/V GEMM (6272 x 9216: 3 rounds of
            // 320 rows against 4 of 256) measured 219 us against 188 us
            constexpr bool allow320 = true;
            const int MB3 = (a.M + 319) / 320;
            const long c256 = (long)((MB2 * NB2 + 255) / 256) * 256, c320 = (long)((MB3 * NB2 + 255) / 256) * 320;
            if (allow320 && c320 * 115 < c256 * 100) {
                const size_t smem3 = (size_t)G2_NS * (16 + 20) * 64 * sizeof(u4);   // 144 KiB
                static DevOnce attr3;
                if (attr3.first()) {
                    hipFuncSetAttribute((const void*)gemm_dma256_k<T, EPI, 10, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
                }
                hipLaunchKernelGGL((gemm_dma256_k<T, EPI, 10, 1>), dim3(MB3 * NB2), dim3(512), smem3, s, a);
                return;
            }
            const size_t smem2 = (size_t)G2_NS * 2 * 16 * 64 * sizeof(u4);   // 128 KiB
            static DevOnce attr2;
            if (attr2.first()) {
                hipFuncSetAttribute((const void*)gemm_dma256_k<T, EPI, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
            }
            hipLaunchKernelGGL((gemm_dma256_k<T, EPI, 8, 1>), dim3(MB2 * NB2), dim3(512), smem2, s, a);
            return;
        }
    }
    const int MB = (a.M + DG_BM - 1) / DG_BM, NB = (a.N + DG_BN - 1) / DG_BN, blocks = MB * NB;
    const int nsteps = a.K / DG_BK;
    int splits = 1;
    if (blocks < 192 && nsteps >= 16 && ws) {
        splits = std::min(std::min((256 + blocks - 1) / blocks, nsteps / 8), 8);
        while (splits > 1 && (size_t)splits * a.M * a.N > ws_floats) --splits;
    }
    const int per = (nsteps + splits - 1) / splits;
    splits = (nsteps + per - 1) / per;
    const size_t smem = (size_t)2 * 2 * 16 * 64 * sizeof(u4);       // 64 KiB
    dim3 grid(blocks, 1, splits), block(256);
    if (splits > 1) {
        static DevOnce attr;
        if (attr.first()) {
            hipFuncSetAttribute((const void*)gemm_dma_k<T, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        }
        hipLaunchKernelGGL((gemm_dma_k<T, EPI, true>), grid, block, smem, s, a, ws, per, ConvGeom(), nullptr);
        const size_t total = (size_t)a.M * (a.N >> 2);
        hipLaunchKernelGGL((splitk_reduce_k<T, EPI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, ws, splits);
    } else {
        static DevOnce attr;
        if (attr.first()) {
            hipFuncSetAttribute((const void*)gemm_dma_k<T, EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        }
        hipLaunchKernelGGL((gemm_dma_k<T, EPI, false>), grid, block, smem, s, a, nullptr, nsteps, ConvGeom(), nullptr);
    }
}

template <typename T>
static void launch_dma_T(const GemmArgs& a, int epi, float* ws, size_t ws_floats, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: launch_dma_epi<T, EPI_NONE>(a, ws, ws_floats, s); break;
        case EPI_RELU: launch_dma_epi<T, EPI_RELU>(a, ws, ws_floats, s); break;
        case EPI_GELU: launch_dma_epi<T, EPI_GELU>(a, ws, ws_floats, s); break;
        case EPI_RESID: launch_dma_epi<T, EPI_RESID>(a, ws, ws_floats, s); break;
        case EPI_RESID_RELU: launch_dma_epi<T, EPI_RESID_RELU>(a, ws, ws_floats, s); break;
        case EPI_SILU_MUL: launch_dma_epi<T, EPI_SILU_MUL>(a, ws, ws_floats, s); break;
        default: break;
    }
}

bool gemm_dma_conv_supported(const GemmArgs& a, const ConvGeom& cg, int epi) {
    return cg.mode == 1 && a.K % DG_BK == 0 && a.N % 16 == 0 && cg.Cin % 8 == 0 && a.M > 32 &&
           (epi == EPI_NONE || epi == EPI_RELU);
}

template <typename T, int EPI>
static void launch_dma_conv_epi(const GemmArgs& a, const ConvGeom& cg, const void* zero16, float* ws, size_t ws_floats, hipStream_t s) {
    const int MB = (a.M + DG_BM - 1) / DG_BM, NB = (a.N + DG_BN - 1) / DG_BN, blocks = MB * NB;
    const size_t smem = (size_t)2 * 2 * 16 * 64 * sizeof(u4);       // 64 KiB
    const int nsteps = a.K / DG_BK;
    // the deep layers of the trunk at batch 1 have 8-25 output tiles and K up to 4608: split K over workgroups
    int splits = 1;
    if (blocks < 192 && nsteps >= 16 && ws && a.N % 4 == 0) {
        splits = std::min(std::min((256 + blocks - 1) / blocks, nsteps / 8), 8);
        while (splits > 1 && (size_t)splits * a.M * a.N > ws_floats) --splits;
    }
    const int per = (nsteps + splits - 1) / splits;
    splits = (nsteps + per - 1) / per;
    static DevOnce attr;
    if (attr.first()) {
        hipFuncSetAttribute((const void*)gemm_dma_k<T, EPI, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)gemm_dma_k<T, EPI, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    if (splits > 1) {
        hipLaunchKernelGGL((gemm_dma_k<T, EPI, true, true>), dim3(blocks, 1, splits), dim3(256), smem, s, a, ws, per, cg, zero16);
        const size_t total = (size_t)a.M * (a.N >> 2);
        hipLaunchKernelGGL((splitk_reduce_k<T, EPI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, ws, splits);
    } else {
        hipLaunchKernelGGL((gemm_dma_k<T, EPI, false, true>), dim3(blocks), dim3(256), smem, s, a, nullptr, nsteps, cg, zero16);
    }
}

void launch_gemm_dma_conv(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, const void* zero16, float* ws, size_t ws_floats,
                          hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, {
        if (epi == EPI_RELU) launch_dma_conv_epi<T, EPI_RELU>(a, cg, zero16, ws, ws_floats, s);
        else launch_dma_conv_epi<T, EPI_NONE>(a, cg, zero16, ws, ws_floats, s);
    });
}

bool gemm_dma_supported(const GemmArgs& a) { return a.K % DG_BK == 0 && a.N % 16 == 0 && a.ldx % 8 == 0 && a.M > 32; }

void launch_gemm_dma(int dtype, const GemmArgs& a, int epi, float* ws, size_t ws_floats, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, launch_dma_T<T>(a, epi, ws, ws_floats, s));
}

}  // namespace rdx

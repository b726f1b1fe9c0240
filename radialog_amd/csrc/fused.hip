// Fused decode attention + o_proj launch (gfx950).
//
// At batch 1 the decode step is a chain of small dependent kernels; attention is latency-bound (32 workgroups, HBM idle)
// and is followed by the o_proj GEMV (33.5 MB of weights, ~8 us alone). Here both run in ONE launch: workgroups
// [0, heads*B) compute attention (attn_body.h, 8 waves), workgroups [heads*B, +N/16) are o_proj tiles (skinny_body.h,
// 8 waves) that put their whole K slice of weights in flight at once and only then wait for the attention output, so the
// weight stream hides under the attention latency.
//
// Hand-off (cdna_hip_programming.md G16, counter form): each attention workgroup stores its 128 outputs (plain stores),
// every storing wave drains vmcnt, __syncthreads, then ONE lane does an agent-scope release fence + asm vmcnt(0) +
// relaxed agent-scope atomic add on the per-layer counter. Each consumer workgroup: ONE lane polls the counter relaxed
// (with s_sleep), then one agent-scope acquire fence, __syncthreads, plain loads. The counter is zeroed once per decode
// step by greedy_step_k (and by prep_prompt_k), never by a consumer.
//
// Residency / deadlock freedom: both roles are 512-thread workgroups capped at 128 VGPRs (half a CU each), so 512
// workgroups fit on the chip; consumers number at most 256 (launcher check), hence producers can always be scheduled,
// and producers never wait on anything. Every spin is bounded: on timeout the consumer sets *err and carries on
// (wrong numbers, reported by the host, never a hang).
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "attn_body.h"
#include "skinny_body.h"

namespace rdx {

typedef __attribute__((address_space(1))) int gint;

struct WaitCounter {
    int* counter; int target; int* err;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0) {
            bool ok = false;
            gint* gc = (gint*)counter;                       // GLOBAL (not flat) agent-scope access, as the G16 recipe requires
            for (int it = 0; it < (1 << 17); ++it) {         // bounded: ~50 ms worst case, then give up loudly
                if (__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) *err = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
};

constexpr int FU_WAVES = 8;

template <typename T, bool XLDS, int MT>
__global__ __launch_bounds__(FU_WAVES * 64, 4) void attn_oproj_k(DecAttnArgs at, GemmArgs g, int n_attn, int* counter, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    if ((int)blockIdx.x < n_attn) {
        const int b = blockIdx.x / at.d.heads, h = blockIdx.x - b * at.d.heads;
        decode_attention_body<T, FU_WAVES>(at, h, b, reinterpret_cast<float*>(fsm));
        // publish: drain this wave's stores, workgroup barrier, one lane releases at agent scope and bumps the counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add((gint*)counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        const int ntiles = gridDim.x - n_attn;
        skinny_tile<T, MT, EPI_RESID, false, FU_WAVES, XLDS>(g, blockIdx.x - n_attn, ntiles, fsm, WaitCounter{counter, n_attn, err});
    }
}

void launch_attn_oproj(int dtype, const DecAttnArgs& a, const GemmArgs& g, int B, int* counter, int* err, hipStream_t s) {
    const int n_attn = a.d.heads * B, ntiles = (g.N + 15) / 16;
    dim3 grid(n_attn + ntiles), block(FU_WAVES * 64);
    const bool xlds = skinny_fits_lds(g.M, g.K);
    const size_t sm_att = decode_attention_smem_floats(FU_WAVES, a.d.max_len) * sizeof(float);
    const size_t sm_gemm = xlds ? (size_t)g.M * g.K * 2 : 0;
    const size_t smem = sm_att > sm_gemm ? sm_att : sm_gemm;
    RDX_DISPATCH_T(dtype, T, {
        if (xlds) hipLaunchKernelGGL((attn_oproj_k<T, true, 1>), grid, block, smem, s, a, g, n_attn, counter, err);
        else if (g.M <= 16) hipLaunchKernelGGL((attn_oproj_k<T, false, 1>), grid, block, smem, s, a, g, n_attn, counter, err);
        else hipLaunchKernelGGL((attn_oproj_k<T, false, 2>), grid, block, smem, s, a, g, n_attn, counter, err);
    });
}

}  // namespace rdx

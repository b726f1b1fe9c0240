// Fused decode attention + o_proj launch (gfx950).
//
// At batch 1 the decode step is a chain of small dependent kernels; attention is latency-bound (32 workgroups, HBM idle)
// and is followed by the o_proj GEMV (33.5 MB of weights, ~8 us alone). Here both run in ONE launch: workgroups
// [0, heads*B) compute attention (attn_body.h, 8 waves), workgroups [heads*B, +N/16) are o_proj tiles (skinny_body.h,
// 8 waves) that put their whole K slice of weights in flight at once and only then wait for the attention output, so the
// weight stream hides under the attention latency.
//
// Hand-off (handoff.h, fence-free form): the attention workgroups store their 128 outputs write-through (8-byte
// agent-scope stores), every storing wave drains vmcnt, __syncthreads, ONE lane bumps a counter shard; each consumer
// polls the 8 shards with 8 lanes (relaxed, s_sleep), __syncthreads, then reads the activation row with 8-byte agent-scope
// loads (L1 bypass). No release / acquire fence on either side (those cost 1.7-6.5 us per workgroup). The counter
// shards are zeroed once per decode step by a memset node.
//
// Residency / deadlock freedom: both roles are 512-thread workgroups capped at 128 VGPRs (half a CU each), so 512
// workgroups fit on the chip; consumers number at most 256 (launcher check), hence producers can always be scheduled,
// and producers never wait on anything. Every spin is bounded: on timeout the consumer sets *err and carries on
// (wrong numbers, reported by the host, never a hang).
#include "rdx_common.h"
#include "rdx_kernels.h"
#include "attn_body.h"
#include "skinny_body.h"
#include "handoff.h"

namespace rdx {

constexpr int FU_WAVES = 8;

template <typename T, bool XLDS, int MT>
__global__ __launch_bounds__(FU_WAVES * 64, 4) void attn_oproj_k(DecAttnArgs at, GemmArgs g, int n_attn, int* counter, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    if ((int)blockIdx.x < n_attn) {
        const int b = blockIdx.x / at.d.heads, h = blockIdx.x - b * at.d.heads;
        decode_attention_body<T, FU_WAVES, true, NoWait, false, 0, true>(at, h, b, reinterpret_cast<float*>(fsm));
        publish_sc1(counter, blockIdx.x);
    } else {
        const int ntiles = gridDim.x - n_attn;
        skinny_tile<T, MT, EPI_RESID, false, FU_WAVES, XLDS, WaitSharded, true>(g, blockIdx.x - n_attn, ntiles, fsm,
                                                                                WaitSharded{counter, n_attn, err, 1, nullptr});
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

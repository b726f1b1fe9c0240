// Workgroup-to-workgroup hand-off inside ONE launch (cdna_hip_programming.md G16, counter form), used by the fused
// attention + o_proj launch and the chained down -> QKV launch (chain.hip).
//
// Producer workgroup: plain stores -> every wave drains vmcnt -> __syncthreads -> ONE lane does an agent-scope release
// fence (L2 write-back: the 8 XCD L2s are not coherent with each other), drains again, then a relaxed agent-scope
// atomic add on the counter. Consumer workgroup: ONE lane polls the counter (relaxed, s_sleep between polls), then an
// agent-scope acquire fence (L1/L2 invalidate), __syncthreads, plain loads.
//
// Liveness: a workgroup only ever waits on workgroups with a SMALLER blockIdx (dispatched no later than itself), and a
// producer never waits on a consumer, so the lowest-indexed unfinished workgroup can always run to completion. Every
// spin is nevertheless bounded: on timeout the consumer sets *err and carries on (wrong numbers, reported by the host
// after the step -- never a hang).
#pragma once
#include "rdx_common.h"

namespace rdx {

typedef __attribute__((address_space(1))) int gint;   // GLOBAL (not flat) address space for the agent-scope accesses

struct NoWait { __device__ __forceinline__ void operator()() const {} };

struct WaitCounter {
    int* counter; int target; int* err;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0) {
            bool ok = false;
            gint* gc = (gint*)counter;
            for (int it = 0; it < (1 << 16); ++it) {         // bounded: tens of ms worst case, then give up loudly
                if (__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (!ok) *err = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
};

// all threads of the workgroup call this after their last store of the published data
__device__ __forceinline__ void publish(int* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add((gint*)counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- fence-free form (write-through payload): used by chain.hip ---------------------------------------------------------
// Payload words are written with relaxed agent-scope 8-byte stores (sc1: write-through, the line leaves the writer's
// L2) and read with relaxed agent-scope 8-byte loads (sc1: bypass the reader's L1), so neither side needs a cache
// fence (a release fence costs 1.7-6.5 us per workgroup, an acquire 1.7 us and more with several workgroups per CU).
// The arrival counter is sharded 8 ways (one 64-byte line each): a single word serialises arrivals at ~13 ns apiece.
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ unsigned long long ld8_agent(const void* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st8_agent(void* p, unsigned long long v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int HO_SHARDS = 8, HO_SHARD_STRIDE = 16;          // ints
constexpr int HO_CTR_INTS = HO_SHARDS * HO_SHARD_STRIDE;

// all threads call this after their last st8_agent of the published data; `idx` = this producer's index in its role
__device__ __forceinline__ void publish_sc1(int* ctr, int idx) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add((gint*)(ctr + (idx & (HO_SHARDS - 1)) * HO_SHARD_STRIDE), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct WaitSharded {     // wait until all `n` producers (indices 0..n-1) of a role have arrived; then read with ld8_agent
    int* ctr; int n; int* err; int naps; long long* tr;   // tr: optional trace slot ([1] = inputs ready); naps: s_sleep(8) repeats between polls (pollers share 8 lines with the arrivals)
    __device__ __forceinline__ void operator()() const {
        if (n <= 0) {
            if (tr && threadIdx.x == 0) tr[1] = (long long)__builtin_amdgcn_s_memrealtime();
            return;
        }
        if (threadIdx.x < HO_SHARDS) {            // 8 lanes, one shard each: ONE 8-request load per poll
            const int k = threadIdx.x;
            const int need = (n - k + HO_SHARDS - 1) / HO_SHARDS;
            gint* gc = (gint*)(ctr + k * HO_SHARD_STRIDE);
            bool ok = false;
            for (int it = 0; it < (1 << 16); ++it) {
                ok = __hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
                if (__all(ok)) break;
                for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(8);
            }
            if (!__all(ok) && threadIdx.x == 0) *err = 1;
            if (tr && threadIdx.x == 0) tr[1] = (long long)__builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
    }
};

}  // namespace rdx

// Beam search support kernels (`num_beams` of test.py:267,:467,:629; `_reorder_cache`,
// modeling_llama_imgemb.py:838-843). The decoder forward of a beam step is the ordinary decode step over batch x beams rows;
// what beam search adds per step is
//   beam_topk_k      log_softmax of every row's logits (rounded to the model dtype, as torch's log_softmax on half logits is)
//                    + the row's running beam score (fp32), then the 2 x beams best (score, flat index) candidates of every batch
//                    group over beams x vocab -- what transformers 4.28.1 beam_search hands to BeamSearchScorer.process;
//   kv_beam_gather_k / kv_beam_scatter_k   `_reorder_cache`: past[:, beam_idx] -- only the GENERATED positions differ between the
//                    beams of a group (they share the prompt), so only those slots are permuted, through a scratch copy;
//   embed_rows_k     the chosen tokens' embedding rows become the next step's input.
// The hypothesis bookkeeping of BeamSearchScorer / BeamHypotheses (tiny, sequential) runs on the host (api.hip: rdx_beam_search).
#include <hip/hip_runtime.h>

#include "rdx_common.h"
#include "rdx_kernels.h"

namespace rdx {

// one workgroup per batch group. logits [groups * beams][vocab] model dtype, beam_scores [groups * beams] fp32.
// cand_score / cand_idx [groups][2 * beams]: descending score; idx = beam * vocab + token. Ties -> lowest flat index.
// logp_out (nullable) [groups * beams][vocab] model dtype: the processed scores HF returns with output_scores=True.
template <typename T>
__global__ __launch_bounds__(1024) void beam_topk_k(const T* __restrict__ logits, const float* __restrict__ beam_scores, int beams, int vocab,
                                                    float* __restrict__ cand_score, int* __restrict__ cand_idx, T* __restrict__ logp_out) {
    __shared__ float red[32];
    __shared__ float s_max[RDX_MAX_BEAMS], s_lse[RDX_MAX_BEAMS];
    __shared__ float sv[1024];
    __shared__ int si[1024];
    __shared__ int chosen[2 * RDX_MAX_BEAMS];
    const int grp = blockIdx.x, tid = threadIdx.x;
    const T* lg = logits + (size_t)grp * beams * vocab;
    for (int b = 0; b < beams; ++b) {
        float m = -INFINITY;
        for (int i = tid; i < vocab; i += blockDim.x) m = fmaxf(m, tof<T>(lg[(size_t)b * vocab + i]));
        m = block_max(m, red);
        float s = 0.f;
        for (int i = tid; i < vocab; i += blockDim.x) s += expf(tof<T>(lg[(size_t)b * vocab + i]) - m);
        s = block_sum(s, red);
        if (tid == 0) { s_max[b] = m; s_lse[b] = logf(s); }
        __syncthreads();
    }
    if (logp_out)
        for (int b = 0; b < beams; ++b)
            for (int i = tid; i < vocab; i += blockDim.x)
                logp_out[((size_t)grp * beams + b) * vocab + i] = fromf<T>((tof<T>(lg[(size_t)b * vocab + i]) - s_max[b]) - s_lse[b]);
    const int total = beams * vocab, want = 2 * beams;
    for (int k = 0; k < want; ++k) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int f = tid; f < total; f += blockDim.x) {
            bool taken = false;
            for (int j = 0; j < k; ++j) taken |= (chosen[j] == f);
            if (taken) continue;
            const int b = f / vocab;
            const float v = rnd<T>((tof<T>(lg[f]) - s_max[b]) - s_lse[b]) + beam_scores[grp * beams + b];
            if (v > bv || (v == bv && f < bi)) { bv = v; bi = f; }
        }
        sv[tid] = bv; si[tid] = bi;
        __syncthreads();
        for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
            if (tid < o) {
                const float v = sv[tid + o];
                const int ix = si[tid + o];
                if (v > sv[tid] || (v == sv[tid] && ix < si[tid])) { sv[tid] = v; si[tid] = ix; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            chosen[k] = si[0];
            cand_score[(size_t)grp * want + k] = sv[0];
            cand_idx[(size_t)grp * want + k] = si[0];
        }
        __syncthreads();
    }
}

void launch_beam_topk(int dtype, const void* logits, const float* beam_scores, int groups, int beams, int vocab, float* cand_score,
                      int* cand_idx, void* logp_out, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((beam_topk_k<T>), dim3(groups), dim3(1024), 0, s, (const T*)logits, beam_scores, beams,
                                                vocab, cand_score, cand_idx, (T*)logp_out));
}

// KV cache [layer][K|V is a separate base][row][head][max_len][128] (2-byte elements); positions [start[r], p1) (multiples of 16, so a K
// slab's 16-position fragment groups move whole) of row `src[r]` -> scratch row r, then scratch -> cache, all layers, K and V.
// start[r] = the first position at which row r's history differs from its new parent's (rounded down to 16): the beams of a prompt share
// long prefixes, and what both rows already hold -- the same tokens run through the same kernels -- is not moved (the round-2 version
// copied every generated position of every re-parented row on every step: O(steps^2) bytes, GBs per step late in a 7B report).
// grid: (rows * heads, layers, 2). 16 bytes per thread per trip.
__global__ __launch_bounds__(256) void kv_beam_copy_k(char* __restrict__ kcache, char* __restrict__ vcache, char* __restrict__ scratch,
                                                      const int* __restrict__ src, const int* __restrict__ start, int rows, int heads, int max_len,
                                                      size_t layer_bytes, int p0, int p1, int to_scratch) {
    const int rh = blockIdx.x, r = rh / heads, h = rh % heads, layer = blockIdx.y, kv = blockIdx.z;
    const int sr = src[r], ps = max(start[r], p0);
    if (sr == r || ps >= p1) return;                       // this row keeps its own history, or shares all of it with its parent
    const size_t stride = (size_t)(p1 - p0) * 256;         // scratch bytes of one (row, head) piece
    const size_t off = (size_t)(ps - p0) * 256, span = (size_t)(p1 - ps) * 256;
    char* base = (kv ? vcache : kcache) + (size_t)layer * layer_bytes;
    char* sc = scratch + ((((size_t)layer * 2 + kv) * rows + r) * heads + h) * stride + off;
    char* cache_src = base + (((size_t)sr * heads + h) * max_len + ps) * 256;
    char* cache_dst = base + (((size_t)r * heads + h) * max_len + ps) * 256;
    const char* from = to_scratch ? cache_src : sc;
    char* to = to_scratch ? sc : cache_dst;
    for (size_t i = (size_t)threadIdx.x * 16; i < span; i += (size_t)blockDim.x * 16) stg16(to + i, ldg16(from + i));
}

void launch_kv_beam_reorder(void* kcache, void* vcache, void* scratch, const int* src, const int* start, int rows, int heads, int layers, int max_len,
                            size_t layer_bytes, int p0, int p1, hipStream_t s) {
    if (p1 <= p0) return;
    dim3 grid(rows * heads, layers, 2), block(256);
    hipLaunchKernelGGL(kv_beam_copy_k, grid, block, 0, s, (char*)kcache, (char*)vcache, (char*)scratch, src, start, rows, heads, max_len, layer_bytes, p0, p1, 1);
    hipLaunchKernelGGL(kv_beam_copy_k, grid, block, 0, s, (char*)kcache, (char*)vcache, (char*)scratch, src, start, rows, heads, max_len, layer_bytes, p0, p1, 0);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_rows_k(const int* __restrict__ tokens, const T* __restrict__ embed, int vocab, T* __restrict__ x, int H) {
    const int b = blockIdx.x;
    int id = tokens[b];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const T* src = embed + (size_t)id * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) stg16(x + (size_t)b * H + i, ldg16(src + i));
}

void launch_embed_rows(int dtype, const int* tokens, const void* embed, int vocab, void* x, int rows, int H, hipStream_t s) {
    RDX_DISPATCH_T(dtype, T, hipLaunchKernelGGL((embed_rows_k<T>), dim3(rows), dim3(256), 0, s, tokens, (const T*)embed, vocab, (T*)x, H));
}

}  // namespace rdx

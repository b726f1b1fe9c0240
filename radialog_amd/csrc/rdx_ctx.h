// librdx internals shared by the api_*.hip translation units: the context, the weight registry and the host-side GEMM dispatch.
// (The public C ABI is include/rdx.h; the kernel launchers are rdx_kernels.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/rdx.h"
#include "rdx_common.h"
#include "rdx_kernels.h"

using namespace rdx;

// decoder rows per context: batch 1-2 chained launches, 3-16 xs16.hip, 3-32 xstat32.hip, 33-128 the row-block family (NB = ceil(rows / 32) row blocks per
// tile walker sharing an XCD's L2: xstat32_k / xsplit32_k<.., BLK>; fp8 weights: the 32-row fp8 x fp8 kernels per block). 128 rows x 512 slots of KV = 68 GB of the 288.
constexpr int RDX_MAX_ROWS = 128;

struct GemmW { void* w = nullptr; int N = 0, K = 0, Npad = 0; void* w8 = nullptr; float* scale = nullptr; };   // w8/scale: fp8 copy
struct RawW { void* p = nullptr; int64_t rows = 0, cols = 0; };

struct LlamaLayer {
    const void *attn_norm, *mlp_norm, *lora_bq, *lora_bv;
    GemmW wqkv, wo, wgu, wdown;
};
struct QLayer {
    GemmW s_wqkv, s_wo, c_wq, c_wo, w1, w2;
    const float *s_bqkv, *s_bo, *s_g, *s_b, *c_bq, *c_bo, *c_g, *c_b, *b1, *b2, *f_g, *f_b;
    int cross_idx;     // -1 = no cross attention in this layer
};
struct VBlock {
    GemmW c1, c2, c3, ds;
    const float *b1, *b2, *b3, *bds;
    bool has_ds;
    int planes, stride;
};

struct PoolBlock {          // one VisionTransformerPooler block (two-image mode)
    const float *n1_g, *n1_b, *n2_g, *n2_b, *bo, *b1, *b2;
    GemmW wqkv, wo, w1, w2;
};

struct GraphKey {
    int B = -1, max_new = 0, eos = 0, pad = 0;
    const void* tokens = nullptr; const void* scores = nullptr;
    bool fixed = false;             // logits always to `scores` itself (beam search) instead of scores + step * stride
    bool operator==(const GraphKey& o) const {
        return B == o.B && max_new == o.max_new && eos == o.eos && pad == o.pad && tokens == o.tokens && scores == o.scores && fixed == o.fixed;
    }
};

struct rdx_ctx {
    rdx_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool finalized = false;
    std::vector<void*> allocs;

    std::map<std::string, GemmW> gemm;
    std::map<std::string, RawW> tens;     // model dtype
    std::map<std::string, RawW> f32;

    // ---- llama ----
    std::vector<LlamaLayer> ll;
    const void *embed = nullptr, *final_norm = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
    GemmW lm_head, img_proj_w;
    const float* img_proj_b = nullptr;
    LlamaDims ld;
    void *kcache = nullptr, *vcache = nullptr;     // [layers][B][heads][max_len][D]
    size_t kv_layer_elems = 0;
    uint8_t* key_mask = nullptr;                   // [B][max_len]
    int *d_img_pos = nullptr, *d_pos_ids = nullptr, *d_pos = nullptr, *d_slot = nullptr, *d_step = nullptr, *d_unf = nullptr;
    float* part_val = nullptr; int* part_idx = nullptr; int n_vtiles = 0;
    // decode-step activations ([max_batch] rows) and prefill activations (grown on demand)
    void *dx = nullptr, *dxn = nullptr, *dqkv = nullptr, *datt = nullptr, *dgu = nullptr;
    float* dxs = nullptr;            // fp8 path, batch 3-128 decode: xscale[rows in 32-row blocks] of the e4m3 activation block(s) in dxn (rmsnorm4096_k<T, 4>)
    void* pxq = nullptr; float* pxs = nullptr;   // fp8 path, prefill: e4m3 activations [rows][max(hidden, inter)] and their scales [rows][4]
    std::string unsupported;         // set by the dispatch when a shape has no kernel in the current mode (fp8 weights); reported by the entry points
    float* kslab = nullptr;          // batch 3-32 decode: fp32 partial slabs [<= 4 groups][32][hidden] of a K-split projection
    int pend_groups = 0;             // launch-time state: slabs written by the last xsplit32 launch, not yet added into dx
    void *px = nullptr, *pxn = nullptr, *pqkv = nullptr, *pq = nullptr, *patt = nullptr, *pgu = nullptr, *pqe = nullptr, *pimg = nullptr;
    size_t prefill_rows = 0;
    float* pslab = nullptr;          // one prompt's prefill (<= 192 rows): fp32 slabs [4][rows][hidden] of the K-split down_proj (xsplit32_k<.., BLK>)
    int cur_B = 0, cur_T = 0, cur_max_new = 0, cur_eos = -1, cur_pad = 0;
    int cur_steps = 0;               // tokens selected since the last prefill (1 after it): bounds rdx_decode_step
    int32_t* cur_tokens = nullptr;
    hipGraphExec_t graph = nullptr;
    bool fuse_attn_oproj = true;     // RDX_FUSE_AO=0 (tests): attention and o_proj as separate launches. Default where supported (batch <= 2):
                                     // ONE 16-wave launch with a fence-free hand-off (chain.hip: attn_oproj16_k)
    bool chain_mlp = true;           // RDX_CHAIN=0 (tests): one kernel per unit. Default at batch <= 2: down(l) -> QKV(l+1) as one chained
                                     // launch, one workgroup per CU (chain.hip: decode_chain_k)
    bool xs16 = true;                // batch 3-16 decode on the one-row-tile family (xs16.hip: norm-prologue projections + un-split o_proj / down, 5 launches
                                     // per layer); RDX_XS16=0 at create / rdx_set_option("xs16", 0): the 32-row family of xstat32.hip (7 launches; A/B leg of the tests)
    bool prompt_blk = true;          // one prompt's K = 4096 projections on xstat32_k<.., BLK> (RDX_PBLK=0 / rdx_set_option("prompt_blk", 0): wstat_k, the A/B leg)
    int chain_naps = 1;              // poll back-off of the chained launch (x s_sleep(8) between polls)
    GemmW cls_fc1, cls_fc2; const float *cls_fc1_b = nullptr, *cls_fc2_b = nullptr;   // findings classifier head
    void *cls_pooled = nullptr, *cls_h = nullptr, *cls_out = nullptr;
    void* zero16 = nullptr;          // 16 zero bytes: source of padding taps in the DMA conv gather
    void* d_cur_rope = nullptr;      // [B][2][128] cos | sin row of each row's current position (written by greedy_step_k)
    ChainLayer* d_clayers = nullptr; int* d_cctr = nullptr;
    int *d_ctr = nullptr, *d_err = nullptr;   // per-layer hand-off counters of the fused launch, sticky error flag
    int flash_min = 512;             // batched causal prefill attention: flash_prefill_k from this many workgroups (RDX_FLASH_MIN at create / rdx_set_option)
    bool pconv_noks = false;         // RDX_PCONV_KSPLIT=0 at create: pconv_k never splits K inside a workgroup (A/B)
    bool trunk_packed = true;        // the ResNet trunk (and the Q-Former GEMMs) on fragment-packed activations (pconv.hip); RDX_PCONV=0 at create: the row-major kernels
    bool ws_ok = false;              // set while the image encoder runs: its many-row GEMMs / convolutions may take wsgemm_k
    float* gemm_ws = nullptr; size_t gemm_ws_floats = 0;      // split-K slabs of gemm_dma_k
    GraphKey gkey;

    // ---- beam search workspaces (rdx_beam_search), sized on first use ----
    void* bm_logits = nullptr; float* bm_scores = nullptr; float* bm_cand_s = nullptr; int* bm_cand_i = nullptr;
    int *bm_tok = nullptr, *bm_src = nullptr; int32_t* bm_out = nullptr; void* bm_scratch = nullptr;
    size_t bm_scratch_bytes = 0; int bm_rows = 0, bm_new = 0;

    void* tf_ws = nullptr; size_t tf_bytes = 0;      // rdx_transform_image: coefficient tables + the horizontal pass's uint8 rows (grown on demand)

    // ---- data-parallel collective (RCCL over xGMI): the one all-gather of generated token ids ----
    void* comm = nullptr; int comm_rank = 0, comm_world = 0;

    // ---- q-former ----
    std::vector<QLayer> ql;
    const void* q_query_ln = nullptr;
    GemmW q_wkv; const float* q_bkv = nullptr; int n_cross = 0;
    // ---- vision ----
    GemmW v_conv1, v_b2v, v_p1, v_p2;
    const float *v_conv1_b = nullptr, *v_p1_b = nullptr, *v_p2_b = nullptr, *v_ln_g = nullptr, *v_ln_b = nullptr;
    std::vector<VBlock> vb;
    std::vector<PoolBlock> pool;                    // optional: present when the pooler weights were uploaded
    const void* pool_emb = nullptr;                 // [2*P][b2v] pos + type embedding (model dtype)
    const float *pool_ng = nullptr, *pool_nb = nullptr, *v_p1f_b = nullptr;
    GemmW v_p1f;                                    // projector conv-1 over the full 2*b2v channels (two-image mode)
    float pool_eps = 1e-6f;
    int enc_batch = 0;
    void *vin = nullptr, *vbuf[4] = {nullptr, nullptr, nullptr, nullptr}, *v_imgemb = nullptr;
    void *qx = nullptr, *qt = nullptr, *qqkv = nullptr, *qctx = nullptr, *qh = nullptr, *qkvx = nullptr;
};

int fail(rdx_ctx* c, int code, const char* fmt, ...);
const char* create_error();

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail((c), -2, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline size_t esz(const rdx_ctx* c) { (void)c; return 2; }

int dalloc(rdx_ctx* c, void** p, size_t bytes);
#define ALLOC(c, ptr, bytes) do { int rc_ = dalloc((c), (void**)&(ptr), (bytes)); if (rc_) return rc_; } while (0)

// release a buffer obtained with ALLOC before it is replaced (workspaces that grow with the batch / prompt length)
template <typename P>
inline void dfree(rdx_ctx* c, P*& p) {
    if (!p) return;
    auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)p);
    if (it != c->allocs.end()) c->allocs.erase(it);
    hipFree((void*)p);
    p = nullptr;
}

// ---- host-side GEMM dispatch (api_dispatch.hip) ----
GemmArgs gargs(const void* X, int ldx, const GemmW& W, const float* bias, void* out, int ldo, int M);
GemmArgs skinny_prenorm(rdx_ctx* c, GemmArgs a, int epi);
void skinny(rdx_ctx* c, GemmArgs a, int epi);
bool down_split_ok(rdx_ctx* c, const LlamaLayer& L, int B);
void launch_ksplit(rdx_ctx* c, const GemmArgs& a);
void launch_down(rdx_ctx* c, const LlamaLayer& L, int B, bool split);
// batch 3-16, model-dtype weights, hidden 4096: the decode step's projections on xs16.hip (no stand-alone RMSNorm, no K-split slabs)
bool blk64_ok(rdx_ctx* c, int B);      // 33-128 rows: the row-block decode family (model-dtype weights, or fp8 x fp8: blk64_fp8)
bool blk64_fp8(rdx_ctx* c);            // ... its fp8 x fp8 form is the one in use (e4m3 decoder weights)
bool xs16_ok(rdx_ctx* c, int B);
void xs16_proj(rdx_ctx* c, GemmArgs a, int epi);          // a.norm_w set, a.X = the row-major residual stream
void xs16_row(rdx_ctx* c, const void* xpacked, const GemmW& W, int B);     // dx += T(xpacked . W^T), in place
void run_gemm(rdx_ctx* c, GemmArgs a, int epi);
void conv_gemm(rdx_ctx* c, const void* X, const GemmW& W, const float* bias, const void* resid, void* out, int B,
               int Hin, int Win, int Cin, int KH, int KW, int stride, int pad, int Hout, int Wout, int epi);
int v_grid(const rdx_config& f);      // side of the trunk's output grid
inline bool fp8_weights(const GemmW& W) { return W.w8 && W.scale && !W.w; }       // the engine's fp8 mode keeps no model-dtype copy
int take_unsupported(rdx_ctx* c);     // nonzero (and rdx_last_error set) when a launch since the last call had no kernel for its shape

// ---- decoder (api_llama.hip) ----
inline void* kv_ptr(rdx_ctx* c, void* base, int layer) { return (char*)base + (size_t)layer * c->kv_layer_elems * 2; }
int prefill_impl(rdx_ctx* c, const int32_t* ids, const int32_t* mask, int B, int T, const float* qformer_embs, int keep,
                 int max_new, int eos_id, int pad_id, int32_t* out_tokens, void* logits);
// evs (timing only, eager launches): a pair of events recorded around every chained down(l) -> QKV(l+1) launch of this step
bool decode_step_launch(rdx_ctx* c, void* logits, const int* out_step, long step_stride, std::vector<hipEvent_t>* evs = nullptr);
int build_graph(rdx_ctx* c, void* scores, bool fixed = false);


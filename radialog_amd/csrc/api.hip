// librdx C ABI: context, weight registry, and the host-side orchestration of the hot path
// (image encode -> Q-Former -> img_proj + <IMG> splice -> Llama prefill -> hipGraph-captured greedy decode).
// Kernels live in gemm.hip / attn.hip / elem.hip; this file only sequences launches on the context's stream.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
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

namespace {

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

}  // namespace

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
    float* kslab = nullptr;          // batch 3-32 decode: fp32 partial slabs [<= 4 groups][32][hidden] of a K-split projection
    int pend_groups = 0;             // launch-time state: slabs written by the last xsplit32 launch, not yet added into dx
    void *px = nullptr, *pxn = nullptr, *pqkv = nullptr, *pq = nullptr, *patt = nullptr, *pgu = nullptr, *pqe = nullptr, *pimg = nullptr;
    size_t prefill_rows = 0;
    int cur_B = 0, cur_T = 0, cur_max_new = 0, cur_eos = -1, cur_pad = 0;
    int cur_steps = 0;               // tokens selected since the last prefill (1 after it): bounds rdx_decode_step
    int32_t* cur_tokens = nullptr;
    hipGraphExec_t graph = nullptr;
    int fuse_attn_oproj = 2;    // RDX_FUSE_AO: attention + o_proj in ONE launch with a fence-free hand-off: 2 = 16-wave kernel (mega.hip,
                                // default where supported: batch <= 2), 1 = 8-wave kernel (fused.hip), 0 = one kernel per unit
    int use_mega = 0;                // RDX_MEGA=n: chained decode-layer kernel (mega.hip), n layers per launch (0 = off, -1 = all)
    int chain_mlp = 2;               // RDX_CHAIN: 2 (default at batch <= 2) = down(l) -> qkv(l+1) as one chained launch (one workgroup per CU);
                                     // 1 = gate/up -> down -> next qkv (measured slower); 0 = one kernel per unit
    int mega_naps = 1;               // RDX_MEGA_NAPS: poll back-off
    int mega_occ = 4;                // RDX_MEGA_OCC: 8 = two workgroups per CU, 4 = one
    GemmW cls_fc1, cls_fc2; const float *cls_fc1_b = nullptr, *cls_fc2_b = nullptr;   // findings classifier head
    void *cls_pooled = nullptr, *cls_h = nullptr, *cls_out = nullptr;
    void* zero16 = nullptr;          // 16 zero bytes: source of padding taps in the DMA conv gather
    void* d_cur_rope = nullptr;      // [B][2][128] cos | sin row of each row's current position (written by greedy_step_k)
    MegaLayer* d_mlayers = nullptr; int* d_mctr = nullptr;
    long long* d_mtrace = nullptr;   // set only during rdx_mega_trace
    int *d_ctr = nullptr, *d_err = nullptr;   // per-layer hand-off counters of the fused launch, sticky error flag
    bool use_dma_gemm = true;        // RDX_DMA=0: route every large-M GEMM through tiled_gemm_k
    bool ws_ok = false;              // set while the image encoder runs: its many-row GEMMs / convolutions may take wsgemm_k
    float* gemm_ws = nullptr; size_t gemm_ws_floats = 0;      // split-K slabs of gemm_dma_k
    GraphKey gkey;

    // ---- beam search workspaces (rdx_beam_search), sized on first use ----
    void* bm_logits = nullptr; float* bm_scores = nullptr; float* bm_cand_s = nullptr; int* bm_cand_i = nullptr;
    int *bm_tok = nullptr, *bm_src = nullptr; int32_t* bm_out = nullptr; void* bm_scratch = nullptr;
    size_t bm_scratch_bytes = 0; int bm_rows = 0, bm_new = 0;

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

static std::string g_create_err;

static int fail(rdx_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail((c), -2, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static size_t esz(const rdx_ctx* c) { (void)c; return 2; }

static int dalloc(rdx_ctx* c, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(c, -3, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    c->allocs.push_back(*p);
    return 0;
}
#define ALLOC(c, ptr, bytes) do { int rc_ = dalloc((c), (void**)&(ptr), (bytes)); if (rc_) return rc_; } while (0)

// release a buffer obtained with ALLOC before it is replaced (workspaces that grow with the batch / prompt length)
template <typename P>
static void dfree(rdx_ctx* c, P*& p) {
    if (!p) return;
    auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)p);
    if (it != c->allocs.end()) c->allocs.erase(it);
    hipFree((void*)p);
    p = nullptr;
}

// ------------------------------------------------------------------------------------------------------------------
// GEMM dispatch
// ------------------------------------------------------------------------------------------------------------------
static GemmArgs gargs(const void* X, int ldx, const GemmW& W, const float* bias, void* out, int ldo, int M) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X; a.ldx = ldx; a.W = W.w; a.bias = bias; a.out = out; a.ldo = ldo;
    a.M = M; a.N = W.N; a.K = W.K; a.n_valid = W.N;
    a.W8 = W.w8; a.wscale = W.scale;
    return a;
}

// Skinny GEMM with an optional fused RMSNorm: fused when the activations fit the kernel's LDS staging path, otherwise
// the rows are normalised once by rmsnorm_k into a scratch buffer (batch-32 decode).
// The RMSNorm of a projection whose rows do not fit the GEMV's LDS stage runs as its own launch in front of it; returns the
// arguments of the GEMM proper (activations = c->dxn)
static GemmArgs skinny_prenorm(rdx_ctx* c, GemmArgs a, int epi) {
    // (batch 3-4 rows would fit the GEMV's LDS stage with the norm fused, but the activation-stationary kernel behind a
    // stand-alone RMSNorm is faster there too: gate/up 41.7 -> 31 + 5 us at batch 4)
    bool standalone = a.norm_w && !skinny_fits_lds(a.M, a.K);
    if (a.norm_w && !standalone && a.M >= xs_min_rows() && c->kslab) {
        GemmArgs t = a;
        t.X = c->dxn; t.ldx = a.K; t.norm_w = nullptr;
        standalone = xstat32_supported(t, epi);
    }
    if (standalone) {
        const void* x = a.X; const void* nw = a.norm_w;
        a.X = c->dxn; a.ldx = a.K; a.norm_w = nullptr;
        // a K-split projection before this one left its residual epilogue to this RMSNorm (xsplit32_k): x += T(sum of slabs)
        const int pend = (x == c->dx) ? c->pend_groups : 0;
        if (pend) c->pend_groups = 0;
        if (xstat32_supported(a, epi)) {       // the normalised rows go straight into the consumer's register-fragment order
            a.xpacked = (a.W8 && a.wscale) ? 2 : 1;
            launch_rmsnorm_packed32(c->cfg.dtype, const_cast<void*>(x), nw, c->dxn, a.M, a.K, a.eps, a.xpacked, pend ? c->kslab : nullptr, pend, c->stream);
        } else if (pend) {
            launch_rmsnorm_packed32(c->cfg.dtype, const_cast<void*>(x), nw, c->dxn, a.M, a.K, a.eps, 0, c->kslab, pend, c->stream);
        } else {
            launch_rmsnorm(c->cfg.dtype, x, nw, c->dxn, a.M, a.K, a.eps, c->stream);
        }
    }
    return a;
}

static void skinny(rdx_ctx* c, GemmArgs a, int epi) {
    launch_skinny_gemm(c->cfg.dtype, skinny_prenorm(c, a, epi), epi, c->stream);
}

// batch 3-32 decode: gate/up (xstat32_k) can hand its SwiGLU output to down_proj fragment-packed, and down_proj then runs
// K-split over 4 workgroups per tile (xsplit32_k), its residual epilogue deferred to the next RMSNorm
static bool down_split_ok(rdx_ctx* c, const LlamaLayer& L, int B) {
    if (B < xs_min_rows() || !c->kslab) return false;
    GemmArgs gu = gargs(c->dxn, c->cfg.hidden, L.wgu, nullptr, c->dgu, c->cfg.inter, B);
    if (!xstat32_supported(gu, EPI_SILU_MUL)) return false;
    GemmArgs dn = gargs(c->dgu, c->cfg.inter, L.wdown, nullptr, c->dx, c->cfg.hidden, B);
    dn.xpacked = (dn.W8 && dn.wscale) ? 2 : 1;       // fp8 weights: the 64-deep fragment order
    return xsplit32_groups(dn) > 0;
}

// A K-split projection (o_proj, down_proj at batch 3-32): its fp32 slabs stay pending for the stand-alone RMSNorm of the
// projection that follows (skinny_prenorm), which adds them, rounds and applies the residual.
static void launch_ksplit(rdx_ctx* c, const GemmArgs& a) {
    launch_xsplit32(c->cfg.dtype, a, c->kslab, c->stream);
    c->pend_groups = xsplit32_groups(a);
}

static void launch_down(rdx_ctx* c, const LlamaLayer& L, int B, bool split) {
    const rdx_config& f = c->cfg;
    GemmArgs a = gargs(c->dgu, f.inter, L.wdown, nullptr, c->dx, f.hidden, B);
    a.resid = c->dx; a.ldr = f.hidden;
    if (split) {
        a.xpacked = (a.W8 && a.wscale) ? 2 : 1;
        launch_ksplit(c, a);
    } else {
        skinny(c, a, EPI_RESID);
    }
}

static void run_gemm(rdx_ctx* c, GemmArgs a, int epi) {
    ConvGeom cg;
    memset(&cg, 0, sizeof(cg));
    if (a.M <= 32) skinny(c, a, epi == EPI_RESID_RELU ? EPI_RESID : epi);
    else if ((c->ws_ok || (a.M > 128 && a.M <= 256 && a.N >= 2048)) && c->zero16 && wsgemm_supported(a, cg, epi))
        launch_wsgemm(c->cfg.dtype, a, cg, epi, c->zero16, c->stream);       // the encoder's GEMMs; a single prompt's prefill GEMMs
    else if (c->use_dma_gemm && gemm_dma_supported(a)) launch_gemm_dma(c->cfg.dtype, a, epi, c->gemm_ws, c->gemm_ws_floats, c->stream);
    else launch_tiled_gemm(c->cfg.dtype, a, cg, epi, c->stream);
}

// ------------------------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------------------------
extern "C" int rdx_create(rdx_ctx** out, int device_id, const rdx_config* cfg) {
    if (!out || !cfg) return fail(nullptr, -1, "rdx_create: null argument");
    if (cfg->dtype != RDX_DTYPE_F16 && cfg->dtype != RDX_DTYPE_BF16) return fail(nullptr, -1, "rdx_create: bad dtype %d", cfg->dtype);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, -2, "rdx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, -1, "rdx_create: device %d out of range (%d devices)", device_id, ndev);
    if (cfg->enable_llama) {
        if (cfg->hidden % cfg->heads || cfg->hidden / cfg->heads != 128)
            return fail(nullptr, -1, "rdx_create: Llama head_dim must be 128 (hidden %d / heads %d)", cfg->hidden, cfg->heads);
        if (cfg->hidden % 32 || cfg->inter % 32 || cfg->inter % 8 || cfg->qformer_dim % 32)
            return fail(nullptr, -1, "rdx_create: hidden/inter/qformer_dim must be multiples of 32");
        if (cfg->max_len % 32 || cfg->max_len <= 0 || cfg->max_len > 1536)
            return fail(nullptr, -1, "rdx_create: max_len must be a multiple of 32 in (0, 1536]");
        if (cfg->max_batch <= 0 || cfg->max_batch > 32) return fail(nullptr, -1, "rdx_create: max_batch must be in [1, 32]");
        if (cfg->lora_r != 0 && cfg->lora_r != 8) return fail(nullptr, -1, "rdx_create: lora_r must be 0 or 8");
    }
    if (cfg->enable_vision) {
        if (cfg->q_hidden / cfg->q_heads != 64 || cfg->q_hidden % cfg->q_heads)
            return fail(nullptr, -1, "rdx_create: Q-Former head_dim must be 64");
    }
    if (cfg->enable_vision || cfg->enable_cls) {
        if (cfg->v_img % 4 || cfg->v_img < 32) return fail(nullptr, -1, "rdx_create: image size must be a multiple of 4, >= 32");
    }
    if (cfg->enable_cls) {
        if (cfg->enable_vision) return fail(nullptr, -1, "rdx_create: a classifier context has enable_vision = 0 (its projector differs)");
        if (cfg->cls_pool <= 0 || cfg->cls_hidden % 32 || cfg->cls_classes <= 0 || cfg->cls_classes > 16)
            return fail(nullptr, -1, "rdx_create: classifier needs pool > 0, hidden %% 32 == 0, 1..16 classes");
    }
    rdx_ctx* c = new rdx_ctx();
    c->cfg = *cfg;
    c->device = device_id;
    if (const char* e = getenv("RDX_FUSE_AO")) c->fuse_attn_oproj = atoi(e);
    if (const char* e = getenv("RDX_MEGA")) c->use_mega = atoi(e);
    if (const char* e = getenv("RDX_CHAIN")) c->chain_mlp = atoi(e);
    if (const char* e = getenv("RDX_MEGA_NAPS")) c->mega_naps = atoi(e);
    if (const char* e = getenv("RDX_MEGA_OCC")) c->mega_occ = atoi(e) == 8 ? 8 : 4;
    if (const char* e = getenv("RDX_DMA")) c->use_dma_gemm = atoi(e) != 0;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, -2, "rdx_create: cannot create stream on device %d", device_id);
    }
    if (hipMalloc(&c->zero16, 64) == hipSuccess) { hipMemset(c->zero16, 0, 64); c->allocs.push_back(c->zero16); } else c->zero16 = nullptr;
    c->gemm_ws_floats = (size_t)16 << 20;          // 64 MiB of fp32 split-K slabs
    if (hipMalloc((void**)&c->gemm_ws, c->gemm_ws_floats * sizeof(float)) != hipSuccess) { c->gemm_ws = nullptr; c->gemm_ws_floats = 0; }
    else c->allocs.push_back(c->gemm_ws);
    *out = c;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time: librccl is only needed by multi-GPU jobs, and the process usually has torch's copy loaded already
// (same soname -> the same instance is shared). No RCCL type crosses the C ABI: the unique id travels as 128 opaque bytes.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, rdx_unique_id, int) = nullptr;       // ncclUniqueId is a 128-byte struct passed by value
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (h) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;       // torch's instance, if it is there
        if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, rdx_unique_id, int))dlsym(h, "ncclCommInitRank");
        AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
        CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) { err = "librccl lacks an expected symbol"; h = nullptr; return false; }
        return true;
    }
    const char* what(int rc) { return GetErrorString ? GetErrorString(rc) : "?"; }
};
Rccl g_rccl;
constexpr int kNcclInt32 = 2;        // ncclInt32 in rccl.h's ncclDataType_t
}  // namespace

extern "C" int rdx_comm_unique_id(rdx_unique_id* id_host) {
    if (!id_host) return fail(nullptr, -1, "rdx_comm_unique_id: null argument");
    if (!g_rccl.load()) return fail(nullptr, -6, "rdx_comm_unique_id: %s", g_rccl.err.c_str());
    const int rc = g_rccl.GetUniqueId(id_host);
    if (rc) return fail(nullptr, -6, "ncclGetUniqueId failed: %s", g_rccl.what(rc));
    return 0;
}

extern "C" int rdx_comm_init(rdx_ctx* c, const rdx_unique_id* id_host, int rank, int world) {
    if (!c || !id_host) return fail(c, -1, "rdx_comm_init: null argument");
    if (world <= 0 || rank < 0 || rank >= world) return fail(c, -1, "rdx_comm_init: bad rank %d / world %d", rank, world);
    if (c->comm) return fail(c, -1, "rdx_comm_init: communicator already initialised");
    if (!g_rccl.load()) return fail(c, -6, "rdx_comm_init: %s", g_rccl.err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = g_rccl.CommInitRank(&c->comm, world, *id_host, rank);
    if (rc) { c->comm = nullptr; return fail(c, -6, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.what(rc)); }
    c->comm_rank = rank; c->comm_world = world;
    return 0;
}

extern "C" int rdx_allgather_tokens(rdx_ctx* c, const int32_t* local, int32_t* global, int rows_local, int n) {
    if (!c || !local || !global || rows_local <= 0 || n <= 0) return fail(c, -1, "rdx_allgather_tokens: bad arguments");
    if (!c->comm) return fail(c, -1, "rdx_allgather_tokens: rdx_comm_init has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = g_rccl.AllGather(local, global, (size_t)rows_local * n, kNcclInt32, c->comm, c->stream);
    if (rc) return fail(c, -6, "ncclAllGather failed: %s", g_rccl.what(rc));
    return 0;
}

extern "C" int rdx_comm_world(rdx_ctx* c) { return (c && c->comm) ? c->comm_world : 0; }

extern "C" void rdx_destroy(rdx_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (c->graph) hipGraphExecDestroy(c->graph);
    for (void* p : c->allocs) hipFree(p);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* rdx_last_error(rdx_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" int rdx_sync(rdx_ctx* c) {
    if (!c) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" void* rdx_stream(rdx_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ------------------------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------------------------
extern "C" int rdx_set_weight(rdx_ctx* c, const char* name, const float* data, int64_t rows, int64_t cols, int kind) {
    if (!c || !name || !data) return fail(c, -1, "rdx_set_weight: null argument");
    if (c->finalized) return fail(c, -1, "rdx_set_weight(%s): weights already finalized", name);
    if (rows <= 0 || cols <= 0) return fail(c, -1, "rdx_set_weight(%s): bad shape [%lld,%lld]", name, (long long)rows, (long long)cols);
    HIPCHK(c, hipSetDevice(c->device));
    const std::string key(name);
    if (kind == RDX_W_GEMM) {
        if (cols % 32) return fail(c, -1, "rdx_set_weight(%s): GEMM K=%lld must be a multiple of 32", name, (long long)cols);
        if (c->gemm.count(key)) return fail(c, -1, "rdx_set_weight(%s): duplicate", name);
        GemmW w;
        w.N = (int)rows; w.K = (int)cols; w.Npad = (int)((rows + 15) / 16 * 16);
        ALLOC(c, w.w, (size_t)w.Npad * w.K * esz(c));
        launch_pack_weight(c->cfg.dtype, data, w.w, w.N, w.K, w.Npad, nullptr, c->stream);
        c->gemm[key] = w;
    } else if (kind == RDX_W_GEMM_FP8) {
        if (cols % 64) return fail(c, -1, "rdx_set_weight(%s): fp8 GEMM K=%lld must be a multiple of 64", name, (long long)cols);
        if (c->gemm.count(key)) return fail(c, -1, "rdx_set_weight(%s): duplicate", name);
        GemmW w;
        w.N = (int)rows; w.K = (int)cols; w.Npad = (int)((rows + 15) / 16 * 16);
        ALLOC(c, w.w, (size_t)w.Npad * w.K * esz(c));            // dequantised copy (prefill, batch > 4)
        ALLOC(c, w.w8, (size_t)w.Npad * w.K);
        ALLOC(c, w.scale, (size_t)w.Npad * sizeof(float));
        launch_pack_weight_fp8(c->cfg.dtype, data, w.w8, w.scale, w.w, w.N, w.K, w.Npad, c->stream);
        c->gemm[key] = w;
    } else if (kind == RDX_W_TENSOR) {
        if (c->tens.count(key)) return fail(c, -1, "rdx_set_weight(%s): duplicate", name);
        RawW r; r.rows = rows; r.cols = cols;
        ALLOC(c, r.p, (size_t)rows * cols * esz(c) + 16);
        launch_from_f32(c->cfg.dtype, data, r.p, (size_t)rows * cols, c->stream);
        c->tens[key] = r;
    } else if (kind == RDX_W_F32) {
        if (c->f32.count(key)) return fail(c, -1, "rdx_set_weight(%s): duplicate", name);
        RawW r; r.rows = rows; r.cols = cols;
        ALLOC(c, r.p, (size_t)rows * cols * sizeof(float));
        HIPCHK(c, hipMemcpyAsync(r.p, data, (size_t)rows * cols * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        c->f32[key] = r;
    } else {
        return fail(c, -1, "rdx_set_weight(%s): unknown kind %d", name, kind);
    }
    // the caller may free `data` as soon as we return
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int v_grid(const rdx_config& f);

namespace {
struct Resolver {
    rdx_ctx* c;
    int rc = 0;
    GemmW g(const std::string& n, int N, int K) {
        auto it = c->gemm.find(n);
        if (it == c->gemm.end()) { if (!rc) rc = fail(c, -4, "missing GEMM weight '%s'", n.c_str()); return GemmW(); }
        if (it->second.N != N || it->second.K != K) {
            if (!rc) rc = fail(c, -4, "weight '%s' has shape [%d,%d], expected [%d,%d]", n.c_str(), it->second.N, it->second.K, N, K);
        }
        return it->second;
    }
    const void* t(const std::string& n, int64_t elems) {
        auto it = c->tens.find(n);
        if (it == c->tens.end()) { if (!rc) rc = fail(c, -4, "missing tensor '%s'", n.c_str()); return nullptr; }
        if (it->second.rows * it->second.cols != elems) { if (!rc) rc = fail(c, -4, "tensor '%s' has %lld elements, expected %lld", n.c_str(), (long long)(it->second.rows * it->second.cols), (long long)elems); }
        return it->second.p;
    }
    const float* f(const std::string& n, int64_t elems) {
        auto it = c->f32.find(n);
        if (it == c->f32.end()) { if (!rc) rc = fail(c, -4, "missing fp32 tensor '%s'", n.c_str()); return nullptr; }
        if (it->second.rows * it->second.cols != elems) { if (!rc) rc = fail(c, -4, "fp32 tensor '%s' has %lld elements, expected %lld", n.c_str(), (long long)(it->second.rows * it->second.cols), (long long)elems); }
        return (const float*)it->second.p;
    }
};
std::string S(const char* fmt, ...) {
    char buf[128];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    return buf;
}
}  // namespace

extern "C" int rdx_finalize_weights(rdx_ctx* c) {
    if (!c) return -1;
    if (c->finalized) return fail(c, -1, "rdx_finalize_weights: already finalized");
    HIPCHK(c, hipSetDevice(c->device));
    const rdx_config& f = c->cfg;
    Resolver R{c};
    if (f.enable_llama) {
        const int H = f.hidden, I = f.inter, r2 = 2 * f.lora_r;
        c->embed = R.t("embed", (int64_t)f.vocab * H);
        c->final_norm = R.t("final_norm", H);
        c->rope_cos = R.t("rope.cos", (int64_t)f.max_pos * 128);
        c->rope_sin = R.t("rope.sin", (int64_t)f.max_pos * 128);
        c->lm_head = R.g("lm_head", f.vocab, H);
        c->img_proj_w = R.g("img_proj.w", H, f.qformer_dim);
        c->img_proj_b = R.f("img_proj.b", H);
        c->ll.resize(f.layers);
        for (int l = 0; l < f.layers; ++l) {
            LlamaLayer& L = c->ll[l];
            L.attn_norm = R.t(S("l%d.attn_norm", l), H);
            L.mlp_norm = R.t(S("l%d.mlp_norm", l), H);
            L.wqkv = R.g(S("l%d.wqkv", l), 3 * H + r2, H);
            L.wo = R.g(S("l%d.wo", l), H, H);
            L.wgu = R.g(S("l%d.wgu", l), 2 * I, H);
            L.wdown = R.g(S("l%d.wdown", l), H, I);
            L.lora_bq = L.lora_bv = nullptr;
            if (f.lora_r) {
                L.lora_bq = R.t(S("l%d.lora_bq", l), (int64_t)H * f.lora_r);
                L.lora_bv = R.t(S("l%d.lora_bv", l), (int64_t)H * f.lora_r);
            }
        }
        if (R.rc) return R.rc;
        c->ld.hidden = H; c->ld.heads = f.heads; c->ld.head_dim = 128; c->ld.qkv_ld = c->ll[0].wqkv.Npad;
        c->ld.lora_r = f.lora_r; c->ld.lora_scale = f.lora_scale; c->ld.max_len = f.max_len; c->ld.max_pos = f.max_pos;
        c->ld.k_perm = (getenv("RDX_KPERM") ? atoi(getenv("RDX_KPERM")) : (f.max_batch * f.heads <= 256)) ? 1 : 0;
        const int B = f.max_batch;
        c->kv_layer_elems = (size_t)B * f.heads * f.max_len * 128;
        ALLOC(c, c->kcache, c->kv_layer_elems * f.layers * 2);
        ALLOC(c, c->vcache, c->kv_layer_elems * f.layers * 2);
        // zero-initialised: decode attention multiplies never-written rows by P = 0 (any finite value is fine, NaN bits are not)
        HIPCHK(c, hipMemset(c->kcache, 0, c->kv_layer_elems * f.layers * 2));
        HIPCHK(c, hipMemset(c->vcache, 0, c->kv_layer_elems * f.layers * 2));
        ALLOC(c, c->key_mask, (size_t)B * f.max_len);
        HIPCHK(c, hipMemset(c->key_mask, 0, (size_t)B * f.max_len));
        ALLOC(c, c->d_img_pos, B * sizeof(int)); ALLOC(c, c->d_pos, B * sizeof(int)); ALLOC(c, c->d_slot, B * sizeof(int));
        ALLOC(c, c->d_step, B * sizeof(int)); ALLOC(c, c->d_unf, B * sizeof(int));
        // hand-off counters: fused attention+o_proj (8 shards x 64 B per layer), then the chained kernel's (mega_ctr_ints)
        // per layer two sharded hand-off counters of 128 ints (fused attention + o_proj at batch <= 2; o_proj / down_proj norm tails
        // at batch 3-32), then the chained kernels' block
        ALLOC(c, c->d_ctr, ((size_t)f.layers * 256 + mega_ctr_ints(f.layers)) * sizeof(int)); ALLOC(c, c->d_err, sizeof(int));
        HIPCHK(c, hipMemset(c->d_ctr, 0, ((size_t)f.layers * 256 + mega_ctr_ints(f.layers)) * sizeof(int)));
        c->d_mctr = c->d_ctr + (size_t)f.layers * 256;
        HIPCHK(c, hipMemset(c->d_err, 0, sizeof(int)));
        {
            std::vector<MegaLayer> ml(f.layers);
            for (int l = 0; l < f.layers; ++l) {
                const LlamaLayer& L = c->ll[l];
                ml[l] = MegaLayer{L.wqkv.w, L.wo.w, L.wgu.w, L.wdown.w, L.attn_norm, L.mlp_norm, L.lora_bq, L.lora_bv,
                                  (char*)c->kcache + (size_t)l * c->kv_layer_elems * 2, (char*)c->vcache + (size_t)l * c->kv_layer_elems * 2,
                                  L.wqkv.w8, L.wo.w8, L.wgu.w8, L.wdown.w8, L.wqkv.scale, L.wo.scale, L.wgu.scale, L.wdown.scale};
            }
            ALLOC(c, c->d_mlayers, ml.size() * sizeof(MegaLayer));
            HIPCHK(c, hipMemcpy(c->d_mlayers, ml.data(), ml.size() * sizeof(MegaLayer), hipMemcpyHostToDevice));
        }
        ALLOC(c, c->d_cur_rope, (size_t)B * 256 * 2);
        ALLOC(c, c->d_pos_ids, (size_t)B * f.max_len * sizeof(int));
        c->n_vtiles = c->lm_head.Npad / 16;
        ALLOC(c, c->part_val, (size_t)B * c->n_vtiles * sizeof(float));
        ALLOC(c, c->part_idx, (size_t)B * c->n_vtiles * sizeof(int));
        ALLOC(c, c->dx, (size_t)B * H * 2); ALLOC(c, c->dxn, (size_t)(B > 2 ? std::max(B, 32) : B) * H * 2);
        if (B > 2) ALLOC(c, c->kslab, (size_t)4 * 32 * H * sizeof(float)); ALLOC(c, c->dqkv, (size_t)B * c->ld.qkv_ld * 2);
        ALLOC(c, c->datt, (size_t)(B > 2 ? std::max(B, 32) : B) * H * 2); ALLOC(c, c->dgu, (size_t)(B > 2 ? std::max(B, 32) : B) * I * 2);
        ALLOC(c, c->pqe, (size_t)B * 32 * f.qformer_dim * 2); ALLOC(c, c->pimg, (size_t)B * 32 * H * 2);      // image splice rows
    }
    if (f.enable_vision) {
        const int H = f.q_hidden, I = f.q_inter;
        c->q_query_ln = R.t("q.query_ln", (int64_t)f.q_nquery * H);
        c->n_cross = 0;
        c->ql.resize(f.q_layers);
        for (int l = 0; l < f.q_layers; ++l) {
            QLayer& L = c->ql[l];
            L.s_wqkv = R.g(S("q%d.self.wqkv", l), 3 * H, H); L.s_bqkv = R.f(S("q%d.self.bqkv", l), 3 * H);
            L.s_wo = R.g(S("q%d.self.wo", l), H, H); L.s_bo = R.f(S("q%d.self.bo", l), H);
            L.s_g = R.f(S("q%d.self.ln_g", l), H); L.s_b = R.f(S("q%d.self.ln_b", l), H);
            L.cross_idx = -1;
            if (l % f.q_cross_freq == 0) {
                L.cross_idx = c->n_cross++;
                L.c_wq = R.g(S("q%d.cross.wq", l), H, H); L.c_bq = R.f(S("q%d.cross.bq", l), H);
                L.c_wo = R.g(S("q%d.cross.wo", l), H, H); L.c_bo = R.f(S("q%d.cross.bo", l), H);
                L.c_g = R.f(S("q%d.cross.ln_g", l), H); L.c_b = R.f(S("q%d.cross.ln_b", l), H);
            }
            L.w1 = R.g(S("q%d.ffn.w1", l), I, H); L.b1 = R.f(S("q%d.ffn.b1", l), I);
            L.w2 = R.g(S("q%d.ffn.w2", l), H, I); L.b2 = R.f(S("q%d.ffn.b2", l), H);
            L.f_g = R.f(S("q%d.ffn.ln_g", l), H); L.f_b = R.f(S("q%d.ffn.ln_b", l), H);
        }
        c->q_wkv = R.g("q.cross.wkv", c->n_cross * 2 * H, f.q_enc_width);
        c->q_bkv = R.f("q.cross.bkv", (int64_t)c->n_cross * 2 * H);
    }
    if (f.enable_vision || f.enable_cls) {
        // vision trunk
        c->v_conv1 = R.g("v.conv1.w", f.v_stem, 7 * 8 * 4); c->v_conv1_b = R.f("v.conv1.b", f.v_stem);
        int cin = f.v_stem;
        for (int li = 0; li < 4; ++li) {
            for (int b = 0; b < f.v_blocks[li]; ++b) {
                VBlock vb;
                vb.planes = f.v_planes[li];
                vb.stride = (b == 0 && li > 0) ? 2 : 1;
                vb.has_ds = (b == 0);
                const std::string p = S("v.l%d.%d.", li + 1, b);
                vb.c1 = R.g(p + "c1.w", vb.planes, cin); vb.b1 = R.f(p + "c1.b", vb.planes);
                vb.c2 = R.g(p + "c2.w", vb.planes, 9 * vb.planes); vb.b2 = R.f(p + "c2.b", vb.planes);
                vb.c3 = R.g(p + "c3.w", 4 * vb.planes, vb.planes); vb.b3 = R.f(p + "c3.b", 4 * vb.planes);
                vb.bds = nullptr;
                if (vb.has_ds) { vb.ds = R.g(p + "ds.w", 4 * vb.planes, cin); vb.bds = R.f(p + "ds.b", 4 * vb.planes); }
                cin = 4 * vb.planes;
                c->vb.push_back(vb);
            }
        }
        c->v_b2v = R.g("v.b2v.w", f.v_b2v, cin);
        c->v_p1 = R.g("v.proj1.w", f.v_proj, f.v_b2v); c->v_p1_b = R.f("v.proj1.b", f.v_proj);
        c->v_p2 = R.g("v.proj2.w", f.v_proj, f.v_proj); c->v_p2_b = R.f("v.proj2.b", f.v_proj);
    }
    if (f.enable_cls) {
        const int gp = v_grid(f) / f.cls_pool;
        if (gp <= 0 || (f.v_proj * gp * gp) % 32) return fail(c, -1, "classifier: pooled feature width %d must be a positive multiple of 32", f.v_proj * gp * gp);
        c->cls_fc1 = R.g("cls.fc1.w", f.cls_hidden, f.v_proj * gp * gp); c->cls_fc1_b = R.f("cls.fc1.b", f.cls_hidden);
        c->cls_fc2 = R.g("cls.fc2.w", f.cls_classes, f.cls_hidden); c->cls_fc2_b = R.f("cls.fc2.b", f.cls_classes);
    }
    if (f.enable_vision) {
        c->v_ln_g = R.f("v.ln.g", f.v_proj); c->v_ln_b = R.f("v.ln.b", f.v_proj);
        if (f.q_enc_width != f.v_proj) return fail(c, -1, "q_enc_width %d != v_proj %d", f.q_enc_width, f.v_proj);
        if (c->tens.count("v.pool.emb")) {          // two-image mode is optional: resolve it only when its weights are there
            const int Cv = f.v_b2v, Pn = v_grid(f) * v_grid(f);
            if (Cv % 32) return fail(c, -1, "ViT pooler needs b2v %% 32 == 0");
            c->pool_emb = R.t("v.pool.emb", (int64_t)2 * Pn * Cv);
            c->pool_ng = R.f("v.pool.norm_g", Cv); c->pool_nb = R.f("v.pool.norm_b", Cv);
            c->v_p1f = R.g("v.proj1f.w", f.v_proj, 2 * Cv); c->v_p1f_b = R.f("v.proj1f.b", f.v_proj);
            for (int i = 0; c->gemm.count(S("v.pool.%d.wqkv", i)); ++i) {
                PoolBlock pb;
                const std::string p = S("v.pool.%d.", i);
                pb.n1_g = R.f(p + "n1_g", Cv); pb.n1_b = R.f(p + "n1_b", Cv); pb.n2_g = R.f(p + "n2_g", Cv); pb.n2_b = R.f(p + "n2_b", Cv);
                pb.wqkv = R.g(p + "wqkv", 3 * Cv, Cv);
                pb.wo = R.g(p + "wo", Cv, Cv); pb.bo = R.f(p + "bo", Cv);
                pb.w1 = R.g(p + "w1", Cv, Cv); pb.b1 = R.f(p + "b1", Cv);
                pb.w2 = R.g(p + "w2", Cv, Cv); pb.b2 = R.f(p + "b2", Cv);
                c->pool.push_back(pb);
            }
        }
    }
    if (R.rc) return R.rc;
    c->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// image encode
// ------------------------------------------------------------------------------------------------------------------
// side of the trunk's output grid: conv1 /2, maxpool /2, then three stride-2 stages of (g - 1) / 2 + 1 (3x3 pad 1 and the
// 1x1 downsample agree): 448 -> 14, 488 -> 16
static int v_grid(const rdx_config& f) {
    int g = f.v_img / 4;
    for (int i = 0; i < 3; ++i) g = (g - 1) / 2 + 1;
    return g;
}

static int ensure_enc_ws(rdx_ctx* c, int B) {
    if (B <= c->enc_batch) return 0;
    const rdx_config& f = c->cfg;
    const int S_ = f.v_img, Hp = S_ + 6;
    // the workspace grows with the largest batch seen: drain the stream, release the old buffers, allocate the new ones
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->enc_batch = 0;
    dfree(c, c->vin);
    for (int i = 0; i < 4; ++i) dfree(c, c->vbuf[i]);
    dfree(c, c->cls_pooled); dfree(c, c->cls_h); dfree(c, c->cls_out);
    dfree(c, c->v_imgemb); dfree(c, c->qx); dfree(c, c->qt); dfree(c, c->qqkv); dfree(c, c->qctx); dfree(c, c->qh); dfree(c, c->qkvx);
    const size_t act = (size_t)B * (S_ / 2) * (S_ / 2) * (size_t)std::max(f.v_stem, 1) * 2;   // conv1 output
    size_t act2 = (size_t)B * (S_ / 4) * (S_ / 4) * (size_t)f.v_planes[0] * 4 * 2;           // layer1 output
    size_t mx = std::max(act, act2);
    const int P = v_grid(f) * v_grid(f);
    mx = std::max(mx, (size_t)B * P * f.v_proj * 2);
    ALLOC(c, c->vin, (size_t)B * Hp * Hp * 4 * 2);
    for (int i = 0; i < 4; ++i) ALLOC(c, c->vbuf[i], mx);
    if (f.enable_cls) {
        ALLOC(c, c->cls_pooled, (size_t)B * c->cls_fc1.K * 2); ALLOC(c, c->cls_h, (size_t)B * f.cls_hidden * 2);
        ALLOC(c, c->cls_out, (size_t)B * 16 * 2 + (size_t)B * f.cls_classes * 2);
    }
    if (!f.enable_vision) { c->enc_batch = B; return 0; }
    ALLOC(c, c->v_imgemb, (size_t)B * P * f.v_proj * 2);
    const size_t M = (size_t)B * f.q_nquery;
    ALLOC(c, c->qx, M * f.q_hidden * 2); ALLOC(c, c->qt, M * f.q_hidden * 2);
    ALLOC(c, c->qqkv, M * 3 * f.q_hidden * 2); ALLOC(c, c->qctx, M * f.q_hidden * 2);
    ALLOC(c, c->qh, M * f.q_inter * 2);
    ALLOC(c, c->qkvx, (size_t)B * P * c->n_cross * 2 * f.q_hidden * 2);
    c->enc_batch = B;
    return 0;
}

static void conv_gemm(rdx_ctx* c, const void* X, const GemmW& W, const float* bias, const void* resid, void* out, int B,
                      int Hin, int Win, int Cin, int KH, int KW, int stride, int pad, int Hout, int Wout, int epi) {
    GemmArgs a = gargs(X, Cin, W, bias, out, W.N, B * Hout * Wout);
    a.resid = resid; a.ldr = W.N;
    ConvGeom cg;
    cg.mode = 1; cg.Hin = Hin; cg.Win = Win; cg.Cin = Cin; cg.Hout = Hout; cg.Wout = Wout;
    cg.KH = KH; cg.KW = KW; cg.stride = stride; cg.pad = pad;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0) cg.mode = 0;
    // memory-bound 1x1 convolutions (K <= 256, tens of thousands of rows): weight-stationary streaming kernel
    if (conv1x1_stream_supported(a, cg, epi)) { launch_conv1x1_stream(c->cfg.dtype, a, cg, epi, c->stream); return; }
    if (c->ws_ok && c->zero16 && wsgemm_supported(a, cg, epi)) { launch_wsgemm(c->cfg.dtype, a, cg, epi, c->zero16, c->stream); return; }
    // 1x1 stride-1 convolutions are plain GEMMs; everything else needs the gather path of the tiled kernel
    if (cg.mode == 0 && c->use_dma_gemm && gemm_dma_supported(a)) launch_gemm_dma(c->cfg.dtype, a, epi, c->gemm_ws, c->gemm_ws_floats, c->stream);
    else if (c->use_dma_gemm && c->zero16 && !resid && gemm_dma_conv_supported(a, cg, epi)) launch_gemm_dma_conv(c->cfg.dtype, a, cg, epi, c->zero16, c->gemm_ws, c->gemm_ws_floats, c->stream);
    else launch_tiled_gemm(c->cfg.dtype, a, cg, epi, c->stream);
}

static int encode_impl(rdx_ctx* c, const float* image, const float* previous, int Bimg, float* qformer_out, float* image_embeds,
                       float* cls_logits = nullptr) {
    if (!c) return -1;
    if (cls_logits) {
        if (!c->finalized || !c->cfg.enable_cls) return fail(c, -1, "rdx_classify_findings: classifier weights not finalized");
        if (!image || Bimg <= 0) return fail(c, -1, "rdx_classify_findings: bad arguments");
    } else {
        if (!c->finalized || !c->cfg.enable_vision) return fail(c, -1, "rdx_encode_image: vision weights not finalized");
        if (!image || !qformer_out || Bimg <= 0) return fail(c, -1, "rdx_encode_image: bad arguments");
    }
    if (previous && !c->pool_emb) return fail(c, -1, "rdx_encode_image2: the ViT-pooler weights (two-image mode) were not loaded");
    HIPCHK(c, hipSetDevice(c->device));
    int B = previous ? 2 * Bimg : Bimg;            // trunk batch: [current ; previous] like torch.cat (encoder.py:119)
    int rc = ensure_enc_ws(c, B);
    if (rc) return rc;
    const rdx_config& f = c->cfg;
    const int dt = f.dtype, S_ = f.v_img, Hp = S_ + 6;
    hipStream_t s = c->stream;
    struct WsScope { rdx_ctx* c; ~WsScope() { c->ws_ok = false; } } ws_scope{c};
    c->ws_ok = true;

    // a1/a2: stem. 7x7/2 conv as implicit GEMM over a zero-padded NHWC4 image: K = 7 x 8(kw, last is zero) x 4(c, last is zero)
    launch_img_prep(dt, image, c->vin, Bimg, S_, 3, Hp, Hp, s);
    if (previous) launch_img_prep(dt, previous, (char*)c->vin + (size_t)Bimg * Hp * Hp * 4 * 2, Bimg, S_, 3, Hp, Hp, s);
    int Hc = S_ / 2;
    if (stem_pool_supported(f.v_stem)) {
        // conv1 + bn1 + relu + maxpool in one launch: the 224^2 x 64 stem output never goes to HBM (stem.hip)
        launch_stem_pool(dt, c->vin, c->v_conv1.w, c->v_conv1_b, c->vbuf[1], B, Hp, Hc, S_ / 4, f.v_stem, s);
    } else {
        conv_gemm(c, c->vin, c->v_conv1, c->v_conv1_b, nullptr, c->vbuf[0], B, Hp, Hp, 4, 7, 8, 2, 0, Hc, Hc, EPI_RELU);
        launch_maxpool(dt, c->vbuf[0], c->vbuf[1], B, Hc, Hc, f.v_stem, s);
    }
    Hc = S_ / 4;
    int C = f.v_stem;
    void *cur = c->vbuf[1], *t1 = c->vbuf[0], *t2 = c->vbuf[2], *t3 = c->vbuf[3];
    for (const VBlock& vb : c->vb) {
        const int Ho = (Hc - 1) / vb.stride + 1;      // 3x3 pad 1 and the 1x1 downsample agree (odd sizes: 61 -> 31)
        conv_gemm(c, cur, vb.c1, vb.b1, nullptr, t1, B, Hc, Hc, C, 1, 1, 1, 0, Hc, Hc, EPI_RELU);
        conv_gemm(c, t1, vb.c2, vb.b2, nullptr, t2, B, Hc, Hc, vb.planes, 3, 3, vb.stride, 1, Ho, Ho, EPI_RELU);
        const void* idt = cur;
        if (vb.has_ds) {
            conv_gemm(c, cur, vb.ds, vb.bds, nullptr, t3, B, Hc, Hc, C, 1, 1, vb.stride, 0, Ho, Ho, EPI_NONE);
            idt = t3;
        }
        conv_gemm(c, t2, vb.c3, vb.b3, idt, t1, B, Ho, Ho, vb.planes, 1, 1, 1, 0, Ho, Ho, EPI_RESID_RELU);
        std::swap(cur, t1);
        Hc = Ho; C = 4 * vb.planes;
    }
    // a3/a4: backbone_to_vit, projector (missing_previous_emb + BN folded into proj1's bias), NHWC output
    const int P = Hc * Hc;
    { GemmArgs a = gargs(cur, C, c->v_b2v, nullptr, t1, f.v_b2v, B * P); run_gemm(c, a, EPI_NONE); }      // [B*P][b2v], NHWC = token order
    const int MP = Bimg * P;
    if (!previous) {
        { GemmArgs a = gargs(t1, f.v_b2v, c->v_p1, c->v_p1_b, t2, f.v_proj, MP); run_gemm(c, a, EPI_RELU); }
    } else {
        // a3': VisionTransformerPooler over [current ; previous] tokens (biovil_t/transformer.py:73-224), then the projector's
        // first conv over the real 2*b2v channels [patch_x | diff_x] (no constant fold in this mode)
        const int Cv = f.v_b2v, L2 = 2 * P, MT = Bimg * L2;
        void *tok = cur, *xe = t2, *qkv = t3;          // the trunk output is dead after backbone_to_vit
        launch_pool_gather(dt, t1, tok, Bimg, P, Cv, s);
        for (const PoolBlock& pb : c->pool) {
            launch_layernorm_ex(dt, tok, Cv, pb.n1_g, pb.n1_b, c->pool_emb, L2, xe, Cv, MT, Cv, c->pool_eps, s);
            { GemmArgs a = gargs(xe, Cv, pb.wqkv, nullptr, qkv, 3 * Cv, MT); run_gemm(c, a, EPI_NONE); }
            AttnArgs at;
            memset(&at, 0, sizeof(at));
            at.Q = qkv; at.K = (const char*)qkv + (size_t)Cv * 2; at.V = (const char*)qkv + (size_t)2 * Cv * 2; at.O = xe;
            at.q_bs = at.k_bs = at.v_bs = (long)L2 * 3 * Cv; at.q_ts = at.k_ts = at.v_ts = 3 * Cv; at.q_hs = at.k_hs = at.v_hs = 32;
            at.o_bs = (long)L2 * Cv; at.o_ts = Cv; at.o_hs = 32;
            at.B = Bimg; at.H = Cv / 32; at.Tq = L2; at.Tk = L2;
            launch_attention(dt, 32, at, s);
            { GemmArgs a = gargs(xe, Cv, pb.wo, pb.bo, tok, Cv, MT); a.resid = tok; a.ldr = Cv; run_gemm(c, a, EPI_RESID); }
            launch_layernorm_ex(dt, tok, Cv, pb.n2_g, pb.n2_b, nullptr, 1, xe, Cv, MT, Cv, c->pool_eps, s);
            { GemmArgs a = gargs(xe, Cv, pb.w1, pb.b1, qkv, Cv, MT); run_gemm(c, a, EPI_GELU); }
            { GemmArgs a = gargs(qkv, Cv, pb.w2, pb.b2, tok, Cv, MT); a.resid = tok; a.ldr = Cv; run_gemm(c, a, EPI_RESID); }
        }
        launch_layernorm_ex(dt, tok, Cv, c->pool_ng, c->pool_nb, nullptr, 1, xe, Cv, MT, Cv, c->pool_eps, s);
        launch_pool_concat(dt, t1, xe, qkv, Bimg, P, Cv, s);                      // [Bimg*P][2*b2v]
        { GemmArgs a = gargs(qkv, 2 * Cv, c->v_p1f, c->v_p1f_b, t2, f.v_proj, MP); run_gemm(c, a, EPI_RELU); }
    }
    { GemmArgs a = gargs(t2, f.v_proj, c->v_p2, c->v_p2_b, t3, f.v_proj, MP); run_gemm(c, a, EPI_NONE); }
    if (cls_logits) {
        // findings classifier head (chexpert_model.py:16-21): avg_pool2d + flatten, fc1 + ReLU, fc2
        const int G = Hc, gp = G / f.cls_pool;
        launch_avgpool_flatten(dt, t3, c->cls_pooled, Bimg, G, f.v_proj, f.cls_pool, s);
        { GemmArgs a = gargs(c->cls_pooled, f.v_proj * gp * gp, c->cls_fc1, c->cls_fc1_b, c->cls_h, f.cls_hidden, Bimg); run_gemm(c, a, EPI_RELU); }
        { GemmArgs a = gargs(c->cls_h, f.cls_hidden, c->cls_fc2, c->cls_fc2_b, c->cls_out, f.cls_classes, Bimg); run_gemm(c, a, EPI_NONE); }
        launch_to_f32(dt, c->cls_out, cls_logits, (size_t)Bimg * f.cls_classes, s);
        HIPCHK(c, hipStreamSynchronize(s));
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    // a5: NCHW reshape scramble + ln_vision
    launch_scramble_layernorm(dt, t3, c->v_ln_g, c->v_ln_b, c->v_imgemb, image_embeds, Bimg, P, f.v_proj, f.v_ln_eps, s);
    B = Bimg;

    // a6: Q-Former, query-only path
    const int H = f.q_hidden, NQ = f.q_nquery, M = B * NQ, KVW = c->n_cross * 2 * H;
    launch_broadcast_rows(dt, c->q_query_ln, c->qx, NQ, H, B, s);
    { GemmArgs a = gargs(c->v_imgemb, f.v_proj, c->q_wkv, c->q_bkv, c->qkvx, KVW, MP); run_gemm(c, a, EPI_NONE); }
    for (const QLayer& L : c->ql) {
        { GemmArgs a = gargs(c->qx, H, L.s_wqkv, L.s_bqkv, c->qqkv, 3 * H, M); run_gemm(c, a, EPI_NONE); }
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.Q = c->qqkv; at.K = (const char*)c->qqkv + (size_t)H * 2; at.V = (const char*)c->qqkv + (size_t)2 * H * 2; at.O = c->qctx;
        at.q_bs = at.k_bs = at.v_bs = (long)NQ * 3 * H; at.q_ts = at.k_ts = at.v_ts = 3 * H; at.q_hs = at.k_hs = at.v_hs = 64;
        at.o_bs = (long)NQ * H; at.o_ts = H; at.o_hs = 64;
        at.B = B; at.H = f.q_heads; at.Tq = NQ; at.Tk = NQ;
        launch_attention(dt, 64, at, s);
        { GemmArgs a = gargs(c->qctx, H, L.s_wo, L.s_bo, c->qt, H, M); a.resid = c->qx; a.ldr = H; run_gemm(c, a, EPI_RESID); }
        launch_layernorm(dt, c->qt, L.s_g, L.s_b, c->qx, nullptr, M, H, f.q_ln_eps, s);
        if (L.cross_idx >= 0) {
            { GemmArgs a = gargs(c->qx, H, L.c_wq, L.c_bq, c->qqkv, H, M); run_gemm(c, a, EPI_NONE); }
            memset(&at, 0, sizeof(at));
            at.Q = c->qqkv; at.q_bs = (long)NQ * H; at.q_ts = H; at.q_hs = 64;
            at.K = (const char*)c->qkvx + (size_t)L.cross_idx * 2 * H * 2; at.V = (const char*)at.K + (size_t)H * 2;
            at.k_bs = at.v_bs = (long)P * KVW; at.k_ts = at.v_ts = KVW; at.k_hs = at.v_hs = 64;
            at.O = c->qctx; at.o_bs = (long)NQ * H; at.o_ts = H; at.o_hs = 64;
            at.B = B; at.H = f.q_heads; at.Tq = NQ; at.Tk = P;
            launch_attention(dt, 64, at, s);
            { GemmArgs a = gargs(c->qctx, H, L.c_wo, L.c_bo, c->qt, H, M); a.resid = c->qx; a.ldr = H; run_gemm(c, a, EPI_RESID); }
            launch_layernorm(dt, c->qt, L.c_g, L.c_b, c->qx, nullptr, M, H, f.q_ln_eps, s);
        }
        { GemmArgs a = gargs(c->qx, H, L.w1, L.b1, c->qh, f.q_inter, M); run_gemm(c, a, EPI_GELU); }
        { GemmArgs a = gargs(c->qh, f.q_inter, L.w2, L.b2, c->qt, H, M); a.resid = c->qx; a.ldr = H; run_gemm(c, a, EPI_RESID); }
        const bool last = (&L == &c->ql.back());
        launch_layernorm(dt, c->qt, L.f_g, L.f_b, c->qx, last ? qformer_out : nullptr, M, H, f.q_ln_eps, s);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int rdx_encode_image(rdx_ctx* c, const float* image, int B, float* qformer_out, float* image_embeds) {
    return encode_impl(c, image, nullptr, B, qformer_out, image_embeds);
}

extern "C" int rdx_classify_findings(rdx_ctx* c, const float* image, int batch, float* logits) {
    if (!logits) return fail(c, -1, "rdx_classify_findings: null output");
    return encode_impl(c, image, nullptr, batch, nullptr, nullptr, logits);
}

extern "C" int rdx_encode_image2(rdx_ctx* c, const float* image, const float* previous_image, int B, float* qformer_out,
                                 float* image_embeds) {
    if (!previous_image) return fail(c, -1, "rdx_encode_image2: previous_image is null (use rdx_encode_image)");
    return encode_impl(c, image, previous_image, B, qformer_out, image_embeds);
}

// ------------------------------------------------------------------------------------------------------------------
// Llama prefill / decode
// ------------------------------------------------------------------------------------------------------------------
static int ensure_prefill_ws(rdx_ctx* c, size_t rows) {
    rows = (rows + 15) & ~(size_t)15;            // the fragment-packed layouts hold whole row tiles of 16
    if (rows <= c->prefill_rows) return 0;
    const rdx_config& f = c->cfg;
    // grows with the largest batch x prompt length seen (test.py-style evaluation: variable prompt lengths): drain the stream,
    // release the old buffers, then allocate; a failure leaves prefill_rows = 0 so the next call starts over
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->prefill_rows = 0;
    dfree(c, c->px); dfree(c, c->pxn); dfree(c, c->pqkv); dfree(c, c->pq); dfree(c, c->patt); dfree(c, c->pgu);
    ALLOC(c, c->px, rows * f.hidden * 2); ALLOC(c, c->pxn, rows * f.hidden * 2);
    ALLOC(c, c->pqkv, rows * c->ld.qkv_ld * 2); ALLOC(c, c->pq, rows * f.hidden * 2);
    ALLOC(c, c->patt, rows * f.hidden * 2); ALLOC(c, c->pgu, rows * f.inter * 2);
    c->prefill_rows = rows;
    return 0;
}

static void* kv_ptr(rdx_ctx* c, void* base, int layer) { return (char*)base + (size_t)layer * c->kv_layer_elems * 2; }

static void lm_head_and_greedy(rdx_ctx* c, const void* x, int B, void* logits, const int* out_step, long step_stride,
                               int advance) {
    const rdx_config& f = c->cfg;
    GemmArgs a = gargs(x, f.hidden, c->lm_head, nullptr, logits, f.vocab, B);
    a.N = c->lm_head.Npad; a.n_valid = f.vocab;
    a.norm_w = c->final_norm; a.eps = f.rms_eps;
    a.part_val = c->part_val; a.part_idx = c->part_idx;
    a.out_step = out_step; a.out_step_stride = step_stride;
    skinny(c, a, EPI_LOGITS);
    launch_greedy_step(f.dtype, c->part_val, c->part_idx, c->n_vtiles, B, c->cur_eos, c->cur_pad, c->cur_max_new,
                       c->cur_tokens, c->d_unf, advance ? c->d_pos : nullptr, advance ? c->d_slot : nullptr, c->d_step,
                       c->embed, f.vocab, c->dx, f.hidden, c->d_pos, c->rope_cos, c->rope_sin, c->d_cur_rope,
                       (c->fuse_attn_oproj || c->use_mega || c->chain_mlp) ? c->d_ctr : nullptr,
                       f.layers * 256 + ((c->use_mega || c->chain_mlp) ? (int)mega_ctr_ints(f.layers) : 0), c->stream);
}

// keep == 0: a fresh prompt. keep > 0: `T` further prompt tokens behind the first `keep` cache slots of the previous call(s)
// (the shared prefix of a multi-turn conversation is not recomputed; no image splice in the continuation).
static int prefill_impl(rdx_ctx* c, const int32_t* ids, const int32_t* mask, int B, int T, const float* qformer_embs, int keep,
                        int max_new, int eos_id, int pad_id, int32_t* out_tokens, void* logits) {
    if (!c) return -1;
    if (!c->finalized || !c->cfg.enable_llama) return fail(c, -1, "rdx_prefill: llama weights not finalized");
    const rdx_config& f = c->cfg;
    if (!ids || !out_tokens || B <= 0 || B > f.max_batch) return fail(c, -1, "rdx_prefill: batch %d outside [1, %d]", B, f.max_batch);
    if (T <= 0 || keep + T + max_new > f.max_len) return fail(c, -1, "rdx_prefill: T (%d) + max_new (%d) exceeds max_len %d", keep + T, max_new, f.max_len);
    if (keep + T + max_new > f.max_pos) return fail(c, -1, "rdx_prefill: sequence exceeds max_position_embeddings %d", f.max_pos);
    if (qformer_embs && T < 32) return fail(c, -1, "rdx_prefill: image splice needs T >= 32");
    HIPCHK(c, hipSetDevice(c->device));
    if (keep > 0) {
        if (B != c->cur_B) return fail(c, -1, "rdx_prefill_append: batch %d differs from the cached conversation's %d", B, c->cur_B);
        std::vector<int> slot(B);
        HIPCHK(c, hipMemcpyAsync(slot.data(), c->d_slot, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int b = 0; b < B; ++b)
            if (keep > slot[b]) return fail(c, -1, "rdx_prefill_append: keep_len %d exceeds the %d cached positions of row %d", keep, slot[b], b);
    }
    const size_t M = (size_t)B * T;
    int rc = ensure_prefill_ws(c, M);
    if (rc) return rc;
    const int dt = f.dtype, H = f.hidden;
    hipStream_t s = c->stream;
    c->cur_B = B; c->cur_T = keep + T; c->cur_max_new = max_new; c->cur_eos = eos_id; c->cur_pad = pad_id; c->cur_tokens = out_tokens;
    c->cur_steps = 1;

    if (keep > 0) {
        launch_prep_append(B, T, keep, c->d_img_pos, c->d_pos_ids, c->d_pos, c->d_slot, c->d_step, c->d_unf, s);
        qformer_embs = nullptr;
    } else {
        launch_prep_prompt(ids, mask, B, T, 32000, pad_id, c->d_img_pos, c->d_pos_ids, c->key_mask, f.max_len, c->d_pos, c->d_slot,
                           c->d_step, c->d_unf, s);
    }
    if (qformer_embs) {
        // a8: img_proj_layer on the model-dtype copy of the Q-Former output (".half()", modeling_llama_imgemb.py:576-579)
        launch_from_f32(dt, qformer_embs, c->pqe, (size_t)B * 32 * f.qformer_dim, s);
        GemmArgs a = gargs(c->pqe, f.qformer_dim, c->img_proj_w, c->img_proj_b, c->pimg, H, B * 32);
        run_gemm(c, a, EPI_NONE);
    }
    launch_embed_splice(dt, ids, c->d_img_pos, c->embed, f.vocab, c->pimg, 32, c->px, B, T, H, qformer_embs ? 1 : 0, s);

    // few rows (one or two prompts): the projections are weight-stream bound -> weight-stationary kernels over fragment-packed
    // activations (wstat.hip); the producers (RMSNorm, attention, the SwiGLU epilogue) write that order directly
    const int mtl = (int)((M + 15) / 16);
    static const int ws_maxm = getenv("RDX_WSTAT_MAXM") ? atoi(getenv("RDX_WSTAT_MAXM")) : 384;
    // measured (tools/prefill_only.py, 32 layers): wstat's time grows with the row tiles of 16, the 128-row tile GEMMs' with the row tiles of 128 --
    // M = 64: 4.84 vs 5.34 ms, 100: 6.06 / 6.26, 160: 7.30 / 7.64, 320: 11.43 / 11.92, but 250: 9.56 / 8.86. Take wstat when the 128-row
    // tiling would pad by 24 rows or more.
    static const int ws_minpad = getenv("RDX_WSTAT_MINPAD") ? atoi(getenv("RDX_WSTAT_MINPAD")) : 24;
    bool ws = (int)M > 32 && (int)M <= ws_maxm && (int)((M + 127) / 128 * 128 - M) >= ws_minpad;
    if (ws) {
        GemmArgs p = gargs(c->pxn, H, c->ll[0].wqkv, nullptr, c->pqkv, c->ld.qkv_ld, (int)M); p.xpacked = 3; p.mtiles = mtl;
        GemmArgs d = gargs(c->pgu, f.inter, c->ll[0].wdown, nullptr, c->px, H, (int)M); d.xpacked = 3; d.mtiles = mtl;
        ws = wstat_supported(p, EPI_NONE) && wstat_supported(d, EPI_RESID);
    }
    auto prompt_gemm = [&](GemmArgs a, int epi, bool packed_out) {
        if (!ws) { run_gemm(c, a, epi); return; }
        a.xpacked = 3; a.mtiles = mtl; a.out_packed = packed_out ? 3 : 0;
        launch_wstat(dt, a, epi, s);
    };
    for (int l = 0; l < f.layers; ++l) {
        const LlamaLayer& L = c->ll[l];
        void* kc = kv_ptr(c, c->kcache, l);
        void* vc = kv_ptr(c, c->vcache, l);
        if (ws) launch_rmsnorm_packed(dt, c->px, L.attn_norm, c->pxn, (int)M, mtl, H, f.rms_eps, s);
        else launch_rmsnorm(dt, c->px, L.attn_norm, c->pxn, (int)M, H, f.rms_eps, s);
        { GemmArgs a = gargs(c->pxn, H, L.wqkv, nullptr, c->pqkv, c->ld.qkv_ld, (int)M); a.N = L.wqkv.Npad; prompt_gemm(a, EPI_NONE, false); }
        // new K/V rows land behind the kept slots
        launch_rope_kv_prefill(dt, c->ld, c->pqkv, L.lora_bq, L.lora_bv, c->rope_cos, c->rope_sin, c->d_pos_ids, c->pq,
                               kc, vc, B, T, keep, s);
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.Q = c->pq; at.q_bs = (long)T * H; at.q_ts = H; at.q_hs = 128;
        at.K = kc; at.V = vc; at.k_bs = at.v_bs = (long)f.heads * f.max_len * 128; at.k_ts = at.v_ts = 128; at.k_hs = at.v_hs = (long)f.max_len * 128;
        at.O = c->patt; at.o_bs = (long)T * H; at.o_ts = H; at.o_hs = 128;
        at.B = B; at.H = f.heads; at.Tq = T; at.Tk = keep + T; at.causal = 1; at.k_perm = c->ld.k_perm; at.key_mask = c->key_mask; at.km_bs = f.max_len;
        at.o_packed_mt = ws ? mtl : 0;
        launch_attention(dt, 128, at, s);
        { GemmArgs a = gargs(c->patt, H, L.wo, nullptr, c->px, H, (int)M); a.resid = c->px; a.ldr = H; prompt_gemm(a, EPI_RESID, false); }
        if (ws) launch_rmsnorm_packed(dt, c->px, L.mlp_norm, c->pxn, (int)M, mtl, H, f.rms_eps, s);
        else launch_rmsnorm(dt, c->px, L.mlp_norm, c->pxn, (int)M, H, f.rms_eps, s);
        { GemmArgs a = gargs(c->pxn, H, L.wgu, nullptr, c->pgu, f.inter, (int)M); prompt_gemm(a, EPI_SILU_MUL, true); }
        { GemmArgs a = gargs(c->pgu, f.inter, L.wdown, nullptr, c->px, H, (int)M); a.resid = c->px; a.ldr = H; prompt_gemm(a, EPI_RESID, false); }
    }
    launch_gather_last(dt, c->px, c->datt, B, T, H, s);      // datt doubles as the [B][H] last-position buffer
    lm_head_and_greedy(c, c->datt, B, logits, nullptr, 0, /*advance=*/0);
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int rdx_prefill(rdx_ctx* c, const int32_t* ids, const int32_t* mask, int B, int T, const float* qformer_embs,
                           int max_new, int eos_id, int pad_id, int32_t* out_tokens, void* logits) {
    return prefill_impl(c, ids, mask, B, T, qformer_embs, 0, max_new, eos_id, pad_id, out_tokens, logits);
}

extern "C" int rdx_prefill_append(rdx_ctx* c, const int32_t* ids_tail, int B, int T_tail, int keep_len, int max_new, int eos_id,
                                  int pad_id, int32_t* out_tokens, void* logits) {
    if (!c) return -1;
    if (keep_len <= 0 || c->cur_B <= 0) return fail(c, -1, "rdx_prefill_append: no cached conversation to continue (keep_len %d)", keep_len);
    return prefill_impl(c, ids_tail, nullptr, B, T_tail, nullptr, keep_len, max_new, eos_id, pad_id, out_tokens, logits);
}

static int decode_loop(rdx_ctx* c, int B, int max_new, int eos_id, void* scores, int* n_steps_host, int use_graph);

// evs (timing only, eager launches): a pair of events recorded around every chained down(l) -> QKV(l+1) launch of this step
static bool decode_step_launch(rdx_ctx* c, void* logits, const int* out_step, long step_stride, std::vector<hipEvent_t>* evs = nullptr) {
    const rdx_config& f = c->cfg;
    const int dt = f.dtype, H = f.hidden, B = c->cur_B;
    hipStream_t s = c->stream;
    if (c->use_mega && mega_supported(c->ld, f.inter, B)) {
        MegaArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.layers = c->d_mlayers; ma.d = c->ld; ma.inter = f.inter; ma.qkv_n = c->ll[0].wqkv.Npad; ma.B = B; ma.eps = f.rms_eps;
        ma.dx = c->dx; ma.dqkv = c->dqkv; ma.datt = c->datt; ma.dgu = c->dgu; ma.cos_t = c->rope_cos; ma.sin_t = c->rope_sin;
        ma.cur_rope = c->d_cur_rope; ma.pos = c->d_pos; ma.slot_b = c->d_slot; ma.key_mask = c->key_mask; ma.ctr = c->d_mctr; ma.err = c->d_err; ma.naps = c->mega_naps; ma.trace = c->d_mtrace;
        const int per = c->use_mega < 0 ? f.layers : c->use_mega;
        for (int l0 = 0; l0 < f.layers; l0 += per) {
            ma.layer0 = l0;
            launch_decode_layers(dt, ma, std::min(per, f.layers - l0), c->mega_occ, s);
        }
        lm_head_and_greedy(c, c->dx, B, logits, out_step, step_stride, /*advance=*/1);
        return false;
    }
    // the hand-off counter shards of the fused launches are cleared by greedy_step_k at the end of the previous step
    // (and of the prefill): a memset node at the head of the step graph was observed to race with the first producers
    // RDX_CHAIN: units chained inside one launch by the fence-free hand-off (mega.hip roles without attention)
    const bool chain = c->chain_mlp && mega_supported(c->ld, f.inter, B) && attn_oproj16_supported(c->ld, f.hidden, f.hidden, B) &&
                       c->fuse_attn_oproj == 2;
    MegaArgs ma;
    if (chain) {
        memset(&ma, 0, sizeof(ma));
        ma.layers = c->d_mlayers; ma.d = c->ld; ma.inter = f.inter; ma.qkv_n = c->ll[0].wqkv.Npad; ma.B = B; ma.eps = f.rms_eps;
        ma.dx = c->dx; ma.dqkv = c->dqkv; ma.datt = c->datt; ma.dgu = c->dgu; ma.cos_t = c->rope_cos; ma.sin_t = c->rope_sin;
        ma.cur_rope = c->d_cur_rope; ma.pos = c->d_pos; ma.slot_b = c->d_slot; ma.key_mask = c->key_mask; ma.ctr = c->d_mctr; ma.err = c->d_err;
        ma.naps = c->mega_naps; ma.trace = nullptr;
        const LlamaLayer& L0 = c->ll[0];        // fp8 weights: the chained roles stream the e4m3 bytes too
        ma.w8 = (L0.wqkv.w8 && L0.wo.w8 && L0.wgu.w8 && L0.wdown.w8 && f.hidden % 64 == 0 && f.inter % 64 == 0) ? 1 : 0;
    }
    for (int l = 0; l < f.layers; ++l) {
        const LlamaLayer& L = c->ll[l];
        if (!chain || l == 0) {
            GemmArgs a = gargs(c->dx, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; a.norm_w = L.attn_norm; a.eps = f.rms_eps;
            skinny(c, a, EPI_NONE);
        }
        DecAttnArgs at;
        at.d = c->ld; at.qkv = c->dqkv; at.lbq = L.lora_bq; at.lbv = L.lora_bv; at.cos_t = c->rope_cos; at.sin_t = c->rope_sin;
        at.pos = c->d_pos; at.slot_b = c->d_slot; at.key_mask = c->key_mask; at.cur_rope = c->d_cur_rope;
        at.kcache = kv_ptr(c, c->kcache, l); at.vcache = kv_ptr(c, c->vcache, l); at.out = c->datt;
        GemmArgs ao = gargs(c->datt, H, L.wo, nullptr, c->dx, H, B);
        ao.resid = c->dx; ao.ldr = H;
        if (c->fuse_attn_oproj == 2 && attn_oproj16_supported(c->ld, L.wo.N, L.wo.K, B)) {
            launch_attn_oproj16(dt, at, ao, B, c->d_ctr + (size_t)l * 256, c->d_err, s);
        } else if (c->fuse_attn_oproj == 1 && (L.wo.N + 15) / 16 <= 256) {
            launch_attn_oproj(dt, at, ao, B, c->d_ctr + (size_t)l * 256, c->d_err, s);
        } else {
            // batch 3-32: attention writes its output fragment-packed and o_proj runs K-split over two workgroups per tile,
            // its residual epilogue folded into the RMSNorm in front of gate/up (xsplit32_k)
            GemmArgs ap = ao; ap.xpacked = (ao.W8 && ao.wscale) ? 2 : 1;
            const int kg = (B >= xs_min_rows() && c->kslab) ? xsplit32_groups(ap) : 0;
            at.out_packed = kg > 0 ? ap.xpacked : 0;
            launch_decode_attention(dt, at, B, s);
            if (kg) launch_ksplit(c, ap);
            else skinny(c, ao, EPI_RESID);
        }
        if (chain && c->chain_mlp == 1) {
            launch_decode_roles(dt, ma, l * 5 + 3, std::min((l + 1) * 5 + 1, f.layers * 5), c->mega_occ, s);
            continue;
        }
        if (chain) {          // RDX_CHAIN=2: gate/up stand-alone, then down(l) -> qkv(l+1) chained
            { GemmArgs a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps; skinny(c, a, EPI_SILU_MUL); }
            if (evs) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s); evs->push_back(e); }
            launch_decode_roles(dt, ma, l * 5 + 4, std::min((l + 1) * 5 + 1, f.layers * 5), c->mega_occ, s);
            if (evs) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s); evs->push_back(e); }
            continue;
        }
        const bool split = down_split_ok(c, L, B);
        { GemmArgs a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps;
          a.out_packed = split ? ((L.wdown.w8 && L.wdown.scale) ? 2 : 1) : 0;
          skinny(c, a, EPI_SILU_MUL); }
        launch_down(c, L, B, split);
    }
    lm_head_and_greedy(c, c->dx, B, logits, out_step, step_stride, /*advance=*/1);
    return chain && c->chain_mlp == 2;
}

// Debug: run ONE eager decode step through the chained decode-layer kernel with per-workgroup timestamps and copy
// them out: host[wg*4 + {0,1,2,3}] = {start, inputs ready, end (100 MHz ticks), role}. 
extern "C" int rdx_mega_trace(rdx_ctx* c, long long* host, int max_wgs) {
    if (!c || !c->finalized || c->cur_B <= 0 || !host) return fail(c, -1, "rdx_mega_trace: run a prefill first");
    if (!c->use_mega || !mega_supported(c->ld, c->cfg.inter, c->cur_B)) return fail(c, -1, "rdx_mega_trace: chained kernel not enabled (RDX_MEGA) or unsupported shape");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)max_wgs * 4 * sizeof(long long);
    HIPCHK(c, hipMalloc(&c->d_mtrace, bytes));
    HIPCHK(c, hipMemsetAsync(c->d_mtrace, 0, bytes, c->stream));
    decode_step_launch(c, nullptr, nullptr, 0);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(host, c->d_mtrace, bytes, hipMemcpyDeviceToHost);
    hipFree(c->d_mtrace);
    c->d_mtrace = nullptr;
    HIPCHK(c, e);
    return 0;
}

// Debug: the stand-alone decode-attention kernel of layer `layer` at the current state, 8 timestamps of workgroup (0,0):
// host[0..6] = after slot load, inputs ready, new token done, barrier 1, scores + barrier, softmax, PV + barrier; [7] = entry.
extern "C" int rdx_attn_trace(rdx_ctx* c, int layer, long long* host) {
    if (!c || !c->finalized || c->cur_B <= 0 || !host) return fail(c, -1, "rdx_attn_trace: run a prefill first");
    HIPCHK(c, hipSetDevice(c->device));
    long long* dtr = nullptr;
    HIPCHK(c, hipMalloc(&dtr, 8 * sizeof(long long)));
    HIPCHK(c, hipMemsetAsync(dtr, 0, 8 * sizeof(long long), c->stream));
    const LlamaLayer& L = c->ll[layer];
    DecAttnArgs at;
    at.d = c->ld; at.qkv = c->dqkv; at.lbq = L.lora_bq; at.lbv = L.lora_bv; at.cos_t = c->rope_cos; at.sin_t = c->rope_sin;
    at.pos = c->d_pos; at.slot_b = c->d_slot; at.key_mask = c->key_mask; at.cur_rope = c->d_cur_rope;
    at.kcache = kv_ptr(c, c->kcache, layer); at.vcache = kv_ptr(c, c->vcache, layer); at.out = c->datt;
    at.trace = dtr;
    launch_decode_attention(c->cfg.dtype, at, c->cur_B, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(host, dtr, 8 * sizeof(long long), hipMemcpyDeviceToHost);
    hipFree(dtr);
    HIPCHK(c, e);
    return 0;
}

// Debug: one stand-alone decode GEMV (what = 1 gate/up, 2 qkv, 4 down, as in rdx_time) of `layer` with per-workgroup
// timestamps: host[tile*8 + {0 entry, 1 weights issued, 2 activations staged, 3 K loop done, 4 all waves done, 5 end}].
extern "C" int rdx_gemv_trace(rdx_ctx* c, int what, int layer, long long* host, int max_tiles) {
    if (!c || !c->finalized || c->cur_B <= 0 || !host) return fail(c, -1, "rdx_gemv_trace: run a prefill first");
    HIPCHK(c, hipSetDevice(c->device));
    const rdx_config& f = c->cfg;
    const int H = f.hidden, B = c->cur_B;
    const LlamaLayer& L = c->ll[layer];
    long long* dtr = nullptr;
    const size_t bytes = (size_t)max_tiles * 8 * sizeof(long long);
    HIPCHK(c, hipMalloc(&dtr, bytes));
    HIPCHK(c, hipMemsetAsync(dtr, 0, bytes, c->stream));
    GemmArgs a;
    if (what == 1) { a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps; }
    else if (what == 2) { a = gargs(c->dx, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; a.norm_w = L.attn_norm; a.eps = f.rms_eps; }
    else { a = gargs(c->dgu, f.inter, L.wdown, nullptr, c->dqkv, H, B); }
    if ((a.N + 15) / 16 > max_tiles) { hipFree(dtr); return fail(c, -1, "rdx_gemv_trace: need room for %d tiles", (a.N + 15) / 16); }
    a.trace = dtr;
    skinny(c, a, what == 1 ? EPI_SILU_MUL : EPI_NONE);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(host, dtr, bytes, hipMemcpyDeviceToHost);
    hipFree(dtr);
    HIPCHK(c, e);
    return 0;
}

extern "C" int rdx_decode_step(rdx_ctx* c, void* logits) {
    if (!c) return -1;
    if (!c->finalized || c->cur_B <= 0) return fail(c, -1, "rdx_decode_step: no prefill has run");
    // every step appends one KV row per batch row and advances the RoPE position: refuse to walk past what the prefill reserved
    if (c->cur_steps >= c->cur_max_new)
        return fail(c, -1, "rdx_decode_step: all %d tokens of this prompt (max_new) have been generated; run a new prefill", c->cur_max_new);
    if (c->cur_T + c->cur_steps > c->cfg.max_len || c->cur_T + c->cur_steps > c->cfg.max_pos)
        return fail(c, -1, "rdx_decode_step: KV cache full (%d prompt + %d generated slots of %d)", c->cur_T, c->cur_steps, c->cfg.max_len);
    HIPCHK(c, hipSetDevice(c->device));
    decode_step_launch(c, logits, nullptr, 0);
    ++c->cur_steps;
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int build_graph(rdx_ctx* c, void* scores, bool fixed = false) {
    const rdx_config& f = c->cfg;
    GraphKey k;
    k.B = c->cur_B; k.max_new = c->cur_max_new; k.eos = c->cur_eos; k.pad = c->cur_pad; k.tokens = c->cur_tokens; k.scores = scores; k.fixed = fixed;
    if (c->graph && k == c->gkey) return 0;
    if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    hipGraph_t g = nullptr;
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    decode_step_launch(c, scores, (scores && !fixed) ? c->d_step : nullptr, (long)c->cur_B * f.vocab);
    HIPCHK(c, hipStreamEndCapture(c->stream, &g));
    HIPCHK(c, hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0));
    HIPCHK(c, hipGraphDestroy(g));
    c->gkey = k;
    return 0;
}

extern "C" int rdx_generate(rdx_ctx* c, const int32_t* ids, const int32_t* mask, int B, int T, const float* qformer_embs,
                            int max_new, int eos_id, int pad_id, int32_t* out_tokens, void* scores, int* n_steps_host,
                            int use_graph) {
    if (!c) return -1;
    if (max_new <= 0) return fail(c, -1, "rdx_generate: max_new must be positive");
    int rc = rdx_prefill(c, ids, mask, B, T, qformer_embs, max_new, eos_id, pad_id, out_tokens, scores);
    if (rc) return rc;
    return decode_loop(c, B, max_new, eos_id, scores, n_steps_host, use_graph);
}

extern "C" int rdx_generate_append(rdx_ctx* c, const int32_t* ids_tail, int B, int T_tail, int keep_len, int max_new, int eos_id,
                                   int pad_id, int32_t* out_tokens, void* scores, int* n_steps_host, int use_graph) {
    if (!c) return -1;
    if (max_new <= 0) return fail(c, -1, "rdx_generate_append: max_new must be positive");
    int rc = rdx_prefill_append(c, ids_tail, B, T_tail, keep_len, max_new, eos_id, pad_id, out_tokens, scores);
    if (rc) return rc;
    return decode_loop(c, B, max_new, eos_id, scores, n_steps_host, use_graph);
}

static int decode_loop(rdx_ctx* c, int B, int max_new, int eos_id, void* scores, int* n_steps_host, int use_graph) {
    int rc = 0;
    const rdx_config& f = c->cfg;
    int done = 1;
    std::vector<int> unf(B, 1);
    auto all_finished = [&]() -> int {
        if (eos_id < 0) return 0;
        if (hipMemcpyAsync(unf.data(), c->d_unf, B * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 0;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
        for (int b = 0; b < B; ++b) if (unf[b]) return 0;
        return 1;
    };
    if (max_new > 1 && !all_finished()) {
        if (use_graph) {
            rc = build_graph(c, scores);
            if (rc) return rc;
        }
        const int check_every = 16;
        while (done < max_new) {
            if (use_graph) {
                HIPCHK(c, hipGraphLaunch(c->graph, c->stream));
            } else {
                void* lg = scores ? (char*)scores + (size_t)done * B * f.vocab * 2 : nullptr;
                decode_step_launch(c, lg, nullptr, 0);
            }
            ++done;
            if (eos_id >= 0 && (done % check_every == 0) && done < max_new && all_finished()) break;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (n_steps_host) *n_steps_host = done;
    c->cur_steps = std::max(done, c->cur_max_new);        // the conversation is complete: further single steps need a new prefill
    int herr = 0;
    HIPCHK(c, hipMemcpy(&herr, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) {
        hipMemset(c->d_err, 0, sizeof(int));
        return fail(c, -5, "rdx_generate: a workgroup hand-off timed out inside a fused/chained launch (results invalid)");
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// beam search: transformers 4.28.1 GenerationMixin.beam_search + BeamSearchScorer, as
// LlamaForCausalLM.generate(num_beams = k) runs it from test.py:467,:629; _reorder_cache = modeling_llama_imgemb.py:838-843
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct BeamHyp { float score; std::vector<int> toks; };
struct BeamHyps {           // transformers 4.28.1 BeamHypotheses
    int num_beams; float length_penalty; int early_stopping;
    std::vector<BeamHyp> beams; float worst = 1e9f;
    void add(const std::vector<int>& gen, int full_len, float sum_logprobs) {
        const float score = sum_logprobs / powf((float)full_len, length_penalty);
        if ((int)beams.size() < num_beams || score > worst) {
            beams.push_back(BeamHyp{score, gen});
            if ((int)beams.size() > num_beams) {
                // sorted([(s, idx)]): drop the lowest score (lowest index on a tie), the runner-up becomes the worst kept score
                int lo = 0;
                for (int i = 1; i < (int)beams.size(); ++i) if (beams[i].score < beams[lo].score) lo = i;
                beams.erase(beams.begin() + lo);
                float w = beams[0].score;
                for (const BeamHyp& h : beams) w = std::min(w, h.score);
                worst = w;
            } else {
                worst = std::min(score, worst);
            }
        }
    }
    bool is_done(float best_sum_logprobs, int cur_len) const {
        if ((int)beams.size() < num_beams) return false;
        if (early_stopping) return true;
        return worst >= best_sum_logprobs / powf((float)cur_len, length_penalty);
    }
};
}  // namespace

extern "C" int rdx_beam_search(rdx_ctx* c, const int32_t* ids, const int32_t* mask, int groups, int num_beams, int T,
                               const float* qformer_embs, int max_new, int eos_id, int pad_id, float length_penalty,
                               int early_stopping, int32_t* out_tokens_host, int32_t* out_len_host, float* out_score_host,
                               void* step_scores, int* n_steps_host) {
    if (!c) return -1;
    if (!c->finalized || !c->cfg.enable_llama) return fail(c, -1, "rdx_beam_search: llama weights not finalized");
    const rdx_config& f = c->cfg;
    const int rows = groups * num_beams;
    if (groups <= 0 || num_beams < 2 || num_beams > RDX_MAX_BEAMS) return fail(c, -1, "rdx_beam_search: num_beams must be in [2, %d]", RDX_MAX_BEAMS);
    if (rows > f.max_batch) return fail(c, -1, "rdx_beam_search: batch %d x %d beams exceeds max_batch %d", groups, num_beams, f.max_batch);
    if (max_new <= 0 || !out_tokens_host || !out_len_host) return fail(c, -1, "rdx_beam_search: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int K2 = 2 * num_beams, V = f.vocab;
    const int p_lo = T / 16 * 16, span = (T + max_new + 15) / 16 * 16 - p_lo;
    if (rows > c->bm_rows || max_new > c->bm_new) {
        HIPCHK(c, hipStreamSynchronize(s));
        dfree(c, c->bm_logits); dfree(c, c->bm_scores); dfree(c, c->bm_cand_s); dfree(c, c->bm_cand_i); dfree(c, c->bm_tok); dfree(c, c->bm_src); dfree(c, c->bm_out);
        const int R = std::max(rows, c->bm_rows), N = std::max(max_new, c->bm_new);
        c->bm_rows = 0;
        ALLOC(c, c->bm_logits, (size_t)R * V * 2); ALLOC(c, c->bm_scores, (size_t)R * 4);
        ALLOC(c, c->bm_cand_s, (size_t)R * 2 * 4); ALLOC(c, c->bm_cand_i, (size_t)R * 2 * 4);
        ALLOC(c, c->bm_tok, (size_t)R * 4); ALLOC(c, c->bm_src, (size_t)R * 4); ALLOC(c, c->bm_out, (size_t)R * N * 4);
        c->bm_rows = R; c->bm_new = N;
    }
    const size_t need = (size_t)f.layers * 2 * rows * f.heads * span * 256;
    if (need > c->bm_scratch_bytes) {
        HIPCHK(c, hipStreamSynchronize(s));
        dfree(c, c->bm_scratch); c->bm_scratch_bytes = 0;
        ALLOC(c, c->bm_scratch, need);
        c->bm_scratch_bytes = need;
    }
    // the prompt: every beam row runs it (HF expands input_ids to batch x beams rows, _expand_inputs_for_generation); EOS handling is
    // the scorer's, so the device-side greedy rule is disabled (eos -1); its argmax tokens go to a dummy buffer and are ignored
    int rc = prefill_impl(c, ids, mask, rows, T, qformer_embs, 0, max_new, -1, pad_id, c->bm_out, c->bm_logits);
    if (rc) return rc;
    rc = build_graph(c, c->bm_logits, /*fixed=*/true);
    if (rc) return rc;

    std::vector<float> beam_scores(rows, -1e9f), cs((size_t)groups * K2);
    for (int g = 0; g < groups; ++g) beam_scores[(size_t)g * num_beams] = 0.f;
    std::vector<int> ci((size_t)groups * K2), next_tok(rows), src(rows);
    std::vector<std::vector<int>> hist(rows), nh(rows);
    std::vector<BeamHyps> hyps(groups);
    for (BeamHyps& h : hyps) { h.num_beams = num_beams; h.length_penalty = length_penalty; h.early_stopping = early_stopping; }
    std::vector<char> done(groups, 0);
    int cur_len = T, steps = 0;
    for (int step = 0; step < max_new; ++step) {
        HIPCHK(c, hipMemcpyAsync(c->bm_scores, beam_scores.data(), rows * sizeof(float), hipMemcpyHostToDevice, s));
        void* lp = step_scores ? (char*)step_scores + (size_t)step * rows * V * 2 : nullptr;
        launch_beam_topk(f.dtype, c->bm_logits, c->bm_scores, groups, num_beams, V, c->bm_cand_s, c->bm_cand_i, lp, s);
        HIPCHK(c, hipMemcpyAsync(cs.data(), c->bm_cand_s, cs.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(ci.data(), c->bm_cand_i, ci.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        ++steps;
        // BeamSearchScorer.process
        for (int g = 0; g < groups; ++g) {
            const int r0 = g * num_beams;
            if (done[g]) {
                for (int b = 0; b < num_beams; ++b) { beam_scores[r0 + b] = 0.f; next_tok[r0 + b] = pad_id; src[r0 + b] = r0 + b; }
                continue;
            }
            int nb = 0;
            for (int rank = 0; rank < K2 && nb < num_beams; ++rank) {
                const float sc = cs[(size_t)g * K2 + rank];
                const int flat = ci[(size_t)g * K2 + rank], from = r0 + flat / V, tok = flat % V;
                if (eos_id >= 0 && tok == eos_id) {
                    if (rank >= num_beams) continue;
                    hyps[g].add(hist[from], cur_len, sc);
                } else {
                    beam_scores[r0 + nb] = sc; next_tok[r0 + nb] = tok; src[r0 + nb] = from;
                    ++nb;
                }
            }
            if (nb < num_beams) return fail(c, -7, "rdx_beam_search: fewer than %d live candidates in group %d (eos-only top-2k)", num_beams, g);
            done[g] = done[g] || hyps[g].is_done(cs[(size_t)g * K2], cur_len);
        }
        for (int r = 0; r < rows; ++r) { nh[r] = hist[src[r]]; nh[r].push_back(next_tok[r]); }
        hist.swap(nh);
        ++cur_len;
        bool all_done = true;
        for (int g = 0; g < groups; ++g) all_done = all_done && done[g];
        if (all_done || cur_len >= T + max_new) break;
        // next forward: _reorder_cache (generated slots only -- the beams of a group share their prompt), chosen tokens in
        HIPCHK(c, hipMemcpyAsync(c->bm_src, src.data(), rows * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->bm_tok, next_tok.data(), rows * sizeof(int), hipMemcpyHostToDevice, s));
        bool identity = true;
        for (int r = 0; r < rows; ++r) identity = identity && src[r] == r;
        if (step > 0 && !identity)
            launch_kv_beam_reorder(c->kcache, c->vcache, c->bm_scratch, c->bm_src, rows, f.heads, f.layers, f.max_len, c->kv_layer_elems * 2,
                                   p_lo, std::min((T + step + 15) / 16 * 16, p_lo + span), s);
        launch_embed_rows(f.dtype, c->bm_tok, c->embed, V, c->dx, rows, f.hidden, s);
        HIPCHK(c, hipGraphLaunch(c->graph, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    c->cur_steps = c->cur_max_new;
    int herr = 0;
    HIPCHK(c, hipMemcpy(&herr, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) { hipMemset(c->d_err, 0, sizeof(int)); return fail(c, -5, "rdx_beam_search: a workgroup hand-off timed out inside a fused launch"); }
    // BeamSearchScorer.finalize: open beams of unfinished groups become hypotheses, the best one per group is returned
    for (int g = 0; g < groups; ++g) {
        if (!done[g]) for (int b = 0; b < num_beams; ++b) hyps[g].add(hist[(size_t)g * num_beams + b], cur_len, beam_scores[(size_t)g * num_beams + b]);
        int best = 0;                     // sorted(..., key = score) is stable and .pop() takes the last: the latest of equal scores
        for (int i = 1; i < (int)hyps[g].beams.size(); ++i) if (hyps[g].beams[i].score >= hyps[g].beams[best].score) best = i;
        const BeamHyp& h = hyps[g].beams[best];
        const int n = std::min((int)h.toks.size(), max_new);
        for (int i = 0; i < max_new; ++i) out_tokens_host[(size_t)g * max_new + i] = i < n ? h.toks[i] : pad_id;
        out_len_host[g] = n;
        if (out_score_host) out_score_host[g] = h.score;
    }
    if (n_steps_host) *n_steps_host = steps;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------------------------------------
extern "C" int rdx_kv_read(rdx_ctx* c, int layer, int which, void* dst) {
    if (!c || !c->finalized || !c->cfg.enable_llama) return fail(c, -1, "rdx_kv_read: no llama state");
    if (layer < 0 || layer >= c->cfg.layers || !dst) return fail(c, -1, "rdx_kv_read: bad arguments");
    if (which) HIPCHK(c, hipMemcpyAsync(dst, kv_ptr(c, c->vcache, layer), c->kv_layer_elems * 2, hipMemcpyDeviceToDevice, c->stream));
    else launch_k_unperm(kv_ptr(c, c->kcache, layer), dst, c->kv_layer_elems / ((size_t)c->cfg.max_len * 128), c->cfg.max_len, c->ld.k_perm, c->stream);   // K: back from the fragment order
    return 0;
}

extern "C" int rdx_hidden_read(rdx_ctx* c, void* dst) {
    if (!c || !c->finalized || !c->cfg.enable_llama || c->cur_B <= 0) return fail(c, -1, "rdx_hidden_read: no llama state");
    HIPCHK(c, hipMemcpyAsync(dst, c->datt, (size_t)c->cur_B * c->cfg.hidden * 2, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int rdx_time(rdx_ctx* c, int what, int iters, float* ms_host) {
    if (!c || !c->finalized || !c->cfg.enable_llama || c->cur_B <= 0) return fail(c, -1, "rdx_time: run a prefill first");
    if (!ms_host || iters <= 0) return fail(c, -1, "rdx_time: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const rdx_config& f = c->cfg;
    const int dt = f.dtype, H = f.hidden, B = c->cur_B;
    const bool same_layer = what >= 10;   // what = 10 + k: unit k on layer 0 only (weights stay cache resident)
    if (same_layer) what -= 10;
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    int launches = 0;
    if (what == 7) {
        // the chained down(l) -> QKV(l+1) launch (decode_layers_k, batch <= 2), IN SITU: `iters` eager decode steps with an event pair
        // around each of its launches (the hand-off counters are only valid inside a real step, so it cannot be looped alone)
        std::vector<int> slot(B);
        HIPCHK(c, hipMemcpyAsync(slot.data(), c->d_slot, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (slot[0] + iters >= f.max_len) return fail(c, -1, "rdx_time: %d steps would overflow the KV cache (slot %d, max_len %d)", iters, slot[0], f.max_len);
        std::vector<hipEvent_t> evs, nul;
        bool chained = true;
        for (int i = 0; i < iters && chained; ++i) {
            chained = decode_step_launch(c, nullptr, nullptr, 0, &evs);
            // calibration: an EMPTY bracket (two event records back to back) costs stream time of its own; it is measured in the same
            // stream, once per step, and subtracted from every bracket below
            hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
            hipEventRecord(a0, c->stream); hipEventRecord(a1, c->stream);
            nul.push_back(a0); nul.push_back(a1);
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->cur_steps = std::max(c->cur_steps, c->cur_max_new);
        double tot = 0.0, empty = 0.0;
        for (size_t i = 0; i + 1 < evs.size(); i += 2) { float m = 0.f; hipEventElapsedTime(&m, evs[i], evs[i + 1]); tot += m; }
        for (size_t i = 0; i + 1 < nul.size(); i += 2) { float m = 0.f; hipEventElapsedTime(&m, nul[i], nul[i + 1]); empty += m; }
        const size_t n = evs.size() / 2;
        // (the empty bracket costs MORE than what an event adds around a kernel -- subtracting it put the result 6 % under rocprof's
        // kernel duration -- so it is measured but NOT subtracted: the bracket = launch gap + kernel, 7 % over rocprof, conservative)
        (void)empty;
        for (hipEvent_t e : evs) hipEventDestroy(e);
        for (hipEvent_t e : nul) hipEventDestroy(e);
        hipEventDestroy(e0); hipEventDestroy(e1);
        if (!chained || n == 0) return fail(c, -1, "rdx_time(7): the chained down -> QKV launch is not active in this configuration (batch > 2, RDX_CHAIN != 2)");
        *ms_host = (float)(tot / (double)n);
        return 0;
    }
    if (what == 0) {
        int rc = build_graph(c, nullptr);
        if (rc) return rc;
        // state advances with every replay: keep the KV slot inside the cache
        std::vector<int> slot(B);
        HIPCHK(c, hipMemcpyAsync(slot.data(), c->d_slot, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (slot[0] + iters >= f.max_len) return fail(c, -1, "rdx_time: %d replays would overflow the KV cache (slot %d, max_len %d)", iters, slot[0], f.max_len);
        HIPCHK(c, hipEventRecord(e0, c->stream));
        for (int i = 0; i < iters; ++i) HIPCHK(c, hipGraphLaunch(c->graph, c->stream));
        HIPCHK(c, hipEventRecord(e1, c->stream));
        launches = iters;
        c->cur_steps = std::max(c->cur_steps, c->cur_max_new);
    } else {
        // projections whose RMSNorm is a launch of its own (rows beyond the GEMV's LDS stage: batch > 4): normalise once,
        // outside the timed region, and time the GEMM launches alone
        int xpk = -1;
        auto pre = [&](GemmArgs a, int epi) {
            if (xpk < 0) { const GemmArgs p = skinny_prenorm(c, a, epi); xpk = p.norm_w ? 0 : (p.X == c->dxn ? 1 + p.xpacked : 0); }
            if (xpk > 0) { a.X = c->dxn; a.ldx = a.K; a.norm_w = nullptr; a.xpacked = xpk - 1; }
            return a;
        };
        if (what == 1) { GemmArgs a = gargs(c->dx, H, c->ll[0].wgu, nullptr, c->dgu, f.inter, B); a.norm_w = c->ll[0].mlp_norm; a.eps = f.rms_eps; pre(a, EPI_SILU_MUL); }
        if (what == 2) { GemmArgs a = gargs(c->dx, H, c->ll[0].wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = c->ll[0].wqkv.Npad; a.norm_w = c->ll[0].attn_norm; a.eps = f.rms_eps; pre(a, EPI_NONE); }
        if (what == 5) { GemmArgs a = gargs(c->dx, H, c->lm_head, nullptr, nullptr, f.vocab, B); a.N = c->lm_head.Npad; a.n_valid = f.vocab; a.norm_w = c->final_norm; a.eps = f.rms_eps; pre(a, EPI_LOGITS); }
        HIPCHK(c, hipEventRecord(e0, c->stream));
        for (int i = 0; i < iters; ++i) {
            if (what == 5) {
                GemmArgs a = gargs(c->dx, H, c->lm_head, nullptr, nullptr, f.vocab, B);
                a.N = c->lm_head.Npad; a.n_valid = f.vocab; a.norm_w = c->final_norm; a.eps = f.rms_eps;
                a.part_val = c->part_val; a.part_idx = c->part_idx;
                launch_skinny_gemm(f.dtype, pre(a, EPI_LOGITS), EPI_LOGITS, c->stream);
                ++launches;
                continue;
            }
            for (int l = 0; l < f.layers; ++l) {
                const LlamaLayer& L = c->ll[same_layer ? 0 : l];
                if (what == 1) { GemmArgs a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps; launch_skinny_gemm(f.dtype, pre(a, EPI_SILU_MUL), EPI_SILU_MUL, c->stream); }
                else if (what == 2) { GemmArgs a = gargs(c->dx, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; a.norm_w = L.attn_norm; a.eps = f.rms_eps; launch_skinny_gemm(f.dtype, pre(a, EPI_NONE), EPI_NONE, c->stream); }
                else if (what == 3) {
                    GemmArgs a = gargs(c->datt, H, L.wo, nullptr, c->dqkv, H, B);
                    GemmArgs ap = a; ap.xpacked = (a.W8 && a.wscale) ? 2 : 1;
                    if (B >= xs_min_rows() && c->kslab && xsplit32_groups(ap)) launch_xsplit32(f.dtype, ap, c->kslab, c->stream);
                    else skinny(c, a, EPI_NONE);
                }
                else if (what == 4) {
                    if (down_split_ok(c, L, B)) { GemmArgs a = gargs(c->dgu, f.inter, L.wdown, nullptr, c->dqkv, H, B); a.xpacked = (a.W8 && a.wscale) ? 2 : 1; launch_xsplit32(f.dtype, a, c->kslab, c->stream); }
                    else { GemmArgs a = gargs(c->dgu, f.inter, L.wdown, nullptr, c->dqkv, H, B); skinny(c, a, EPI_NONE); }
                }
                else if (what == 6) {   // decode attention at the current slot (re-appends the same KV row: idempotent)
                    DecAttnArgs at;
                    at.d = c->ld; at.qkv = c->dqkv; at.lbq = L.lora_bq; at.lbv = L.lora_bv; at.cos_t = c->rope_cos; at.sin_t = c->rope_sin;
                    at.pos = c->d_pos; at.slot_b = c->d_slot; at.key_mask = c->key_mask; at.cur_rope = c->d_cur_rope;
                    at.kcache = kv_ptr(c, c->kcache, same_layer ? 0 : l); at.vcache = kv_ptr(c, c->vcache, same_layer ? 0 : l); at.out = c->datt;
                    launch_decode_attention(dt, at, B, c->stream);
                }
                else return fail(c, -1, "rdx_time: unknown unit %d", what);
                ++launches;
            }
        }
        HIPCHK(c, hipEventRecord(e1, c->stream));
    }
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_host = ms / (float)launches;
    return 0;
}


// Kernel benchmark hook: `iters` launches of one GEMM (ksize = 0: out[M][N] = epi(X[M][K] W^T), M = rows) or one NHWC convolution
// (ksize = 1 / 3: batch `rows` images of H x H x K channels -> N channels, `stride`, pad = ksize / 2) through the SAME dispatch
// the encoder / prefill use (run_gemm / conv_gemm), timed with HIP events on the context's stream. Contents are zeros; epi 3 / 6
// read a residual. Returns ms per launch.
extern "C" int rdx_kernel_bench(rdx_ctx* c, int rows, int N, int K, int H, int ksize, int stride, int epi, int iters, float* ms_host,
                                long long* trace_host, int trace_wgs) {
    if (!c || !ms_host || iters <= 0 || rows <= 0) return fail(c, -1, "rdx_kernel_bench: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const int Kg = ksize ? ksize * ksize * K : K;
    if (Kg % 32 || N % 16) return fail(c, -1, "rdx_kernel_bench: need K %% 32 == 0 and N %% 16 == 0");
    const int Ho = ksize ? (H + 2 * (ksize / 2) - ksize) / stride + 1 : 0;
    const size_t M = ksize ? (size_t)rows * Ho * Ho : (size_t)rows, Min = ksize ? (size_t)rows * H * H : (size_t)rows;
    char* buf = nullptr;
    const bool kb_wstat = getenv("RDX_KB_WSTAT") && atoi(getenv("RDX_KB_WSTAT")) && !ksize;
    const size_t xb = ((Min + 15) & ~(size_t)15) * K * 2 + 64, wb = (size_t)N * Kg * 2, ob = M * N * 2 + 64, bb = (size_t)N * 4;
    HIPCHK(c, hipMalloc((void**)&buf, xb + wb + 2 * ob + bb));
    HIPCHK(c, hipMemsetAsync(buf, 0, xb + wb + 2 * ob + bb, c->stream));
    GemmW w; w.N = N; w.K = Kg; w.Npad = N; w.w = buf + xb;
    void *out = buf + xb + wb, *res = buf + xb + wb + ob;
    const float* bias = (const float*)(buf + xb + wb + 2 * ob);
    const bool need_res = epi == EPI_RESID || epi == EPI_RESID_RELU;
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    struct WsScope { rdx_ctx* c; ~WsScope() { c->ws_ok = false; } } ws_scope{c};
    c->ws_ok = true;
    long long* dtr = nullptr;
    if (trace_host && trace_wgs > 0) { HIPCHK(c, hipMalloc((void**)&dtr, (size_t)trace_wgs * 8 * sizeof(long long))); HIPCHK(c, hipMemset(dtr, 0, (size_t)trace_wgs * 64)); }
    auto once = [&]() {
        if (ksize) conv_gemm(c, buf, w, bias, need_res ? res : nullptr, out, rows, H, H, K, ksize, ksize, stride, ksize / 2, Ho, Ho, epi);
        else {
            GemmArgs a = gargs(buf, K, w, bias, out, N, (int)M); a.resid = need_res ? res : nullptr; a.ldr = N; a.trace = dtr;
            if (kb_wstat) {       // RDX_KB_WSTAT=1: the single prompt's weight-stationary kernel on (zero) fragment-packed activations
                a.xpacked = 3; a.mtiles = (int)((M + 15) / 16); a.bias = nullptr;
                if (wstat_supported(a, epi)) { launch_wstat(c->cfg.dtype, a, epi, c->stream); return; }
            }
            run_gemm(c, a, epi);
        }
    };
    once();
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) once();
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (dtr) { hipMemcpy(trace_host, dtr, (size_t)trace_wgs * 64, hipMemcpyDeviceToHost); hipFree(dtr); }
    hipFree(buf);
    HIPCHK(c, hipGetLastError());
    *ms_host = ms / (float)iters;
    return 0;
}

// Microbenchmark: GB/s that `wgs` workgroups (256 threads) pull from a cache-resident buffer, `bytes_per_wg` each (shared = 1: all
// read the same region), read `reps` times; mode 0 = global_load_dwordx4, 1 = global_load_lds_dwordx4. Returns the aggregate GB/s.
extern "C" int rdx_l2_bench(rdx_ctx* c, int mode, long long bytes_per_wg, int shared, int reps, int wgs, float* gbps_host) {
    if (!c || !gbps_host || bytes_per_wg < 65536 || wgs <= 0 || reps <= 0) return fail(c, -1, "rdx_l2_bench: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    char* buf = nullptr;
    const size_t total = shared ? (size_t)bytes_per_wg : (size_t)bytes_per_wg * wgs;
    HIPCHK(c, hipMalloc((void**)&buf, total + 64));
    HIPCHK(c, hipMemsetAsync(buf, 1, total + 64, c->stream));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    launch_l2_bench(mode, buf, (size_t)bytes_per_wg, shared, 1, wgs, (unsigned*)(buf + total), c->stream);        // warm the caches
    HIPCHK(c, hipEventRecord(e0, c->stream));
    launch_l2_bench(mode, buf, (size_t)bytes_per_wg, shared, reps, wgs, (unsigned*)(buf + total), c->stream);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(buf);
    HIPCHK(c, hipGetLastError());
    *gbps_host = (float)((double)bytes_per_wg * wgs * reps / (ms * 1e-3) / 1e9);
    return 0;
}

// One bare GEMM through the production kernels (unit tests / kernel benchmarks): out = epilogue(X . W^T).
// X, resid, norm_w, out are model-dtype device tensors; W [N][K] and bias [N] are fp32 device tensors (W is packed here).
extern "C" int rdx_gemm_test(rdx_ctx* c, const void* X, const float* W, const float* bias, const void* resid, void* out,
                             int M, int N, int K, int epi, const void* norm_w, float eps, int force) {
    if (!c || !X || !W || !out) return fail(c, -1, "rdx_gemm_test: null argument");
    if (K % 32 || N % 16) return fail(c, -1, "rdx_gemm_test: need K %% 32 == 0 and N %% 16 == 0");
    HIPCHK(c, hipSetDevice(c->device));
    GemmW w;
    w.N = N; w.K = K; w.Npad = N;
    void* wp = nullptr;
    HIPCHK(c, hipMalloc(&wp, (size_t)N * K * 2 + (force == 4 ? (size_t)N * K + (size_t)N * 4 : 0)));
    w.w = wp;
    if (force == 4) {                     // fp8 weights: quantise here, stream the e4m3 bytes
        if (K % 64) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: fp8 needs K %% 64 == 0"); }
        w.w8 = (char*)wp + (size_t)N * K * 2;
        w.scale = (float*)((char*)wp + (size_t)N * K * 3);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, wp, N, K, N, c->stream);
        force = 1;
    } else
    launch_pack_weight(c->cfg.dtype, W, wp, N, K, N, nullptr, c->stream);
    GemmArgs a = gargs(X, K, w, bias, out, epi == EPI_SILU_MUL ? N / 2 : N, M);
    a.resid = resid; a.ldr = N;
    a.norm_w = norm_w; a.eps = eps;
    void* xn = nullptr;
    bool split8 = false;
    if (force == 6) {       // force 5 with fp8 weights
        if (K % 64) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: fp8 needs K %% 64 == 0"); }
        hipFree(wp);
        HIPCHK(c, hipMalloc(&wp, (size_t)N * K * 3 + (size_t)N * 4));
        w.w = wp; w.w8 = (char*)wp + (size_t)N * K * 2; w.scale = (float*)((char*)wp + (size_t)N * K * 3);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, wp, N, K, N, c->stream);
        a = gargs(X, K, w, bias, out, N, M);
        a.resid = resid; a.ldr = N; a.eps = eps;
        split8 = true; force = 5;
    }
    if (force == 5) {       // K-split slab path: pack X -> xsplit32_k -> slab combine (+ residual) at the launch boundary; out = resid + T(X W^T)
        if (epi != EPI_RESID || !resid || M <= 16 || M > 32) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: force 5 needs epi 3 and 16 < M <= 32"); }
        char* tmp = nullptr;
        const size_t xb = (size_t)32 * K * 2, sb = (size_t)4 * 32 * N * 4;
        HIPCHK(c, hipMalloc((void**)&tmp, 2 * xb + sb));
        launch_rmsnorm_packed32(c->cfg.dtype, const_cast<void*>(X), nullptr, tmp, M, K, eps, split8 ? 2 : 1, nullptr, 0, c->stream);   // w = null: re-layout only
        a.X = tmp; a.xpacked = split8 ? 2 : 1; a.norm_w = nullptr;
        const int kg = xsplit32_groups(a);
        if (!kg) { hipFree(wp); hipFree(tmp); return fail(c, -1, "rdx_gemm_test: shape not supported by xsplit32_k"); }
        launch_xsplit32(c->cfg.dtype, a, (float*)(tmp + 2 * xb), c->stream);
        HIPCHK(c, hipMemcpyAsync(out, resid, (size_t)M * N * 2, hipMemcpyDeviceToDevice, c->stream));
        launch_rmsnorm_packed32(c->cfg.dtype, out, nullptr, tmp + xb, M, N, eps, 0, (const float*)(tmp + 2 * xb), kg, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        hipFree(wp); hipFree(tmp);
        return 0;
    }
    if (force == 8) {         // the single prompt's weight-stationary kernel (wstat.hip): RMSNorm / re-layout into the fragment-packed order, then the GEMM
        const int mtl = (M + 15) / 16;
        char* tmp = nullptr;
        HIPCHK(c, hipMalloc((void**)&tmp, (size_t)mtl * 16 * K * 2));
        launch_rmsnorm_packed(c->cfg.dtype, X, norm_w, tmp, M, mtl, K, eps, c->stream);        // norm_w == null: re-layout only
        a.X = tmp; a.xpacked = 3; a.mtiles = mtl; a.norm_w = nullptr;
        if (!wstat_supported(a, epi)) { hipFree(wp); hipFree(tmp); return fail(c, -1, "rdx_gemm_test: shape not supported by wstat_k"); }
        launch_wstat(c->cfg.dtype, a, epi, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        hipFree(wp); hipFree(tmp);
        return 0;
    }
    if (force == 7) {         // the encoder's many-row kernel (wsgemm.hip), tile shape via RDX_WS_CFG
        ConvGeom cg0;
        memset(&cg0, 0, sizeof(cg0));
        a.norm_w = nullptr;
        if (!c->zero16 || a.K % 64 || a.N % 16) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: shape not supported by wsgemm_k"); }
        launch_wsgemm(c->cfg.dtype, a, cg0, epi, c->zero16, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        hipFree(wp);
        return 0;
    }
    const bool use_skinny = force == 1 || (force == 0 && M <= 32);
    if (use_skinny) {
        if (M > 32) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: skinny path needs M <= 32"); }
        void* keep = c->dxn;                       // skinny() pre-normalises into c->dxn when the rows do not fit the LDS
        if (a.norm_w && !skinny_fits_lds(M, K)) {
            HIPCHK(c, hipMalloc(&xn, (size_t)32 * K * 2));
            c->dxn = xn;
        }
        skinny(c, a, epi == EPI_RESID_RELU ? EPI_RESID : epi);
        c->dxn = keep;
    } else {
        if (a.norm_w) {
            HIPCHK(c, hipMalloc(&xn, (size_t)M * K * 2));
            launch_rmsnorm(c->cfg.dtype, X, norm_w, xn, M, K, eps, c->stream);
            a.X = xn; a.norm_w = nullptr;
        }
        ConvGeom cg;
        memset(&cg, 0, sizeof(cg));
        if (force == 3 || (force == 0 && c->use_dma_gemm && gemm_dma_supported(a))) {
            if (!gemm_dma_supported(a)) { hipFree(wp); if (xn) hipFree(xn); return fail(c, -1, "rdx_gemm_test: shape not supported by gemm_dma_k"); }
            launch_gemm_dma(c->cfg.dtype, a, epi, c->gemm_ws, c->gemm_ws_floats, c->stream);
        } else {
            launch_tiled_gemm(c->cfg.dtype, a, cg, epi, c->stream);
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    hipFree(wp);
    if (xn) hipFree(xn);
    return 0;
}

// the lm_head epilogue (logits + per-tile argmax partials) of the weight-streaming kernels on a bare GEMM: what the decode
// step runs before greedy_step_k; the partials are reduced here on the host with the same tie rule (lowest index)
extern "C" int rdx_logits_test(rdx_ctx* c, const void* X, const float* W, int M, int N, int n_valid, int K, void* out_logits,
                               int32_t* argmax_host, int fp8) {
    if (!c || !X || !W || !out_logits || !argmax_host) return fail(c, -1, "rdx_logits_test: null argument");
    if (K % 32 || N % 16 || M > 32 || n_valid > N || (fp8 && K % 64)) return fail(c, -1, "rdx_logits_test: bad shape");
    HIPCHK(c, hipSetDevice(c->device));
    GemmW w;
    w.N = N; w.K = K; w.Npad = N;
    const int nt = N / 16;
    char* wp = nullptr;
    const size_t wb = (size_t)N * K * 2, qb = fp8 ? (size_t)N * K + (size_t)N * 4 : 0, pb = (size_t)M * nt * 4;
    HIPCHK(c, hipMalloc((void**)&wp, wb + qb + 2 * pb + (size_t)M * K * 2));
    w.w = wp;
    if (fp8) {
        w.w8 = wp + wb; w.scale = (float*)(wp + wb + (size_t)N * K);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, wp, N, K, N, c->stream);
    } else {
        launch_pack_weight(c->cfg.dtype, W, wp, N, K, N, nullptr, c->stream);
    }
    GemmArgs a = gargs(X, K, w, nullptr, out_logits, N, M);
    a.n_valid = n_valid;
    a.part_val = (float*)(wp + wb + qb); a.part_idx = (int*)(wp + wb + qb + pb);
    launch_skinny_gemm(c->cfg.dtype, a, EPI_LOGITS, c->stream);
    std::vector<float> pv((size_t)M * nt);
    std::vector<int> pi((size_t)M * nt);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(pv.data(), a.part_val, pb, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(pi.data(), a.part_idx, pb, hipMemcpyDeviceToHost));
    hipFree(wp);
    for (int m = 0; m < M; ++m) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int t = 0; t < nt; ++t) {
            const float v = pv[(size_t)m * nt + t]; const int ix = pi[(size_t)m * nt + t];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
        argmax_host[m] = bi;
    }
    return 0;
}

// librdx C ABI, part 1: context lifecycle and the weight registry (rdx_create .. rdx_finalize_weights).
// The hot path itself is sequenced in api_encode.hip (image encode), api_llama.hip (prefill / decode / beam search) and
// api_comm.hip (the RCCL all-gather); api_debug.hip holds the introspection / benchmark hooks. Kernels live in the other .hip files.
#include "rdx_ctx.h"

static std::string g_create_err;

int fail(rdx_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

const char* create_error() { return g_create_err.c_str(); }

int dalloc(rdx_ctx* c, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(c, -3, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    c->allocs.push_back(*p);
    return 0;
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
        if (cfg->max_batch <= 0 || cfg->max_batch > RDX_MAX_ROWS) return fail(nullptr, -1, "rdx_create: max_batch must be in [1, %d]", RDX_MAX_ROWS);
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
    if (const char* e = getenv("RDX_FUSE_AO")) c->fuse_attn_oproj = atoi(e) != 0;
    if (const char* e = getenv("RDX_CHAIN")) c->chain_mlp = atoi(e) != 0;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, -2, "rdx_create: cannot create stream on device %d", device_id);
    }
    { const char* e = getenv("RDX_FLASH_MIN"); if (e) c->flash_min = atoi(e); }
    { const char* e = getenv("RDX_XS16"); if (e) c->xs16 = atoi(e) != 0; }
    { const char* e = getenv("RDX_PBLK"); if (e) c->prompt_blk = atoi(e) != 0; }
    { const char* e = getenv("RDX_PCONV_KSPLIT"); c->pconv_noks = e && atoi(e) == 0; }
    { const char* e = getenv("RDX_PCONV"); c->trunk_packed = !(e && atoi(e) == 0); }      // read once (A/B legs of the tests set it before rdx_create)
    if (hipMalloc(&c->zero16, 64) == hipSuccess) { hipMemset(c->zero16, 0, 64); c->allocs.push_back(c->zero16); } else c->zero16 = nullptr;
    c->gemm_ws_floats = (size_t)16 << 20;          // 64 MiB of fp32 split-K slabs
    if (hipMalloc((void**)&c->gemm_ws, c->gemm_ws_floats * sizeof(float)) != hipSuccess) { c->gemm_ws = nullptr; c->gemm_ws_floats = 0; }
    else c->allocs.push_back(c->gemm_ws);
    *out = c;
    return 0;
}

void rdx_comm_release(rdx_ctx* c);      // api_comm.hip

extern "C" void rdx_destroy(rdx_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    rdx_comm_release(c);
    if (c->graph) hipGraphExecDestroy(c->graph);
    for (void* p : c->allocs) hipFree(p);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* rdx_last_error(rdx_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

// radialog_amd/build.py source_hash() of the kernel / ABI sources this library was compiled from (round 6: _lib.load() refuses a library whose
// hash is not the tree's -- a stale .so cannot be tested or benchmarked silently)
#ifndef RDX_BUILD_HASH
#define RDX_BUILD_HASH "unstamped"
#endif
extern "C" const char* rdx_build_hash(void) { return RDX_BUILD_HASH; }

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
        ALLOC(c, w.w8, (size_t)w.Npad * w.K);                     // the ONLY copy: e4m3 bytes + one scale per row (w.w stays null)
        ALLOC(c, w.scale, (size_t)w.Npad * sizeof(float));
        launch_pack_weight_fp8(c->cfg.dtype, data, w.w8, w.scale, nullptr, w.N, w.K, w.Npad, c->stream);
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

// A source tensor in fp16 / bf16 (a real checkpoint is stored that way) is widened on the device, one tensor at a time, and takes the
// fp32 path above: bit-identical to uploading the widened tensor, without a 2x host copy of the whole checkpoint.
extern "C" int rdx_set_weight_typed(rdx_ctx* c, const char* name, const void* data, int src_dtype, int64_t rows, int64_t cols, int kind) {
    if (src_dtype == RDX_SRC_F32) return rdx_set_weight(c, name, (const float*)data, rows, cols, kind);
    if (!c || !name || !data) return fail(c, -1, "rdx_set_weight_typed: null argument");
    if (src_dtype != RDX_SRC_F16 && src_dtype != RDX_SRC_BF16) return fail(c, -1, "rdx_set_weight_typed(%s): unknown source dtype %d", name, src_dtype);
    if (rows <= 0 || cols <= 0) return fail(c, -1, "rdx_set_weight_typed(%s): bad shape [%lld,%lld]", name, (long long)rows, (long long)cols);
    HIPCHK(c, hipSetDevice(c->device));
    float* wide = nullptr;
    const size_t n = (size_t)rows * cols;
    hipError_t e = hipMalloc((void**)&wide, n * sizeof(float));
    if (e != hipSuccess) return fail(c, -3, "rdx_set_weight_typed(%s): hipMalloc(%zu bytes) failed: %s", name, n * sizeof(float), hipGetErrorString(e));
    launch_to_f32(src_dtype == RDX_SRC_F16 ? DT_F16 : DT_BF16, data, wide, n, c->stream);
    const int rc = rdx_set_weight(c, name, wide, rows, cols, kind);        // synchronises the stream before it returns
    hipFree(wide);
    return rc;
}

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
        // hand-off counters (zero at the start of every step: greedy_step_k clears them): per layer 256 ints for the fused attention +
        // o_proj launch (8 shards x one 64-byte line), then the chained down -> QKV launches' block (chain_ctr_ints)
        ALLOC(c, c->d_ctr, ((size_t)f.layers * 256 + chain_ctr_ints(f.layers)) * sizeof(int)); ALLOC(c, c->d_err, sizeof(int));
        HIPCHK(c, hipMemset(c->d_ctr, 0, ((size_t)f.layers * 256 + chain_ctr_ints(f.layers)) * sizeof(int)));
        c->d_cctr = c->d_ctr + (size_t)f.layers * 256;
        HIPCHK(c, hipMemset(c->d_err, 0, sizeof(int)));
        {
            std::vector<ChainLayer> cl(f.layers);
            for (int l = 0; l < f.layers; ++l) {
                const LlamaLayer& L = c->ll[l];
                cl[l] = ChainLayer{L.wqkv.w, L.wdown.w, L.attn_norm, L.wqkv.w8, L.wdown.w8, L.wqkv.scale, L.wdown.scale};
            }
            ALLOC(c, c->d_clayers, cl.size() * sizeof(ChainLayer));
            HIPCHK(c, hipMemcpy(c->d_clayers, cl.data(), cl.size() * sizeof(ChainLayer), hipMemcpyHostToDevice));
        }
        ALLOC(c, c->d_cur_rope, (size_t)B * 256 * 2);
        ALLOC(c, c->d_pos_ids, (size_t)B * f.max_len * sizeof(int));
        c->n_vtiles = c->lm_head.Npad / 16;
        ALLOC(c, c->part_val, (size_t)B * c->n_vtiles * sizeof(float));
        ALLOC(c, c->part_idx, (size_t)B * c->n_vtiles * sizeof(int));
        // decode activations between kernels: from batch 3 whole fragment-packed blocks -- 32 rows (xstat32.hip / xs16.hip), or the row tiles of 33-128 rows
        const int Bp = B > 2 ? (B + 31) / 32 * 32 : B;            // whole 32-row blocks (the fp8 row-block kernels index their last block in full)
        ALLOC(c, c->dx, (size_t)B * H * 2); ALLOC(c, c->dxn, (size_t)Bp * H * 2);
        if (B > 2) ALLOC(c, c->kslab, (size_t)4 * Bp * H * sizeof(float));        // fp32 slabs of a K-split projection: [<= 4 groups][32, or the row tiles of 33-128 rows][H]
        ALLOC(c, c->dqkv, (size_t)B * c->ld.qkv_ld * 2);
        ALLOC(c, c->dxs, (size_t)std::max(32, Bp) * sizeof(float));
        ALLOC(c, c->datt, (size_t)Bp * H * 2); ALLOC(c, c->dgu, (size_t)Bp * I * 2);
        ALLOC(c, c->pqe, (size_t)B * 32 * f.qformer_dim * 2); ALLOC(c, c->pimg, (size_t)B * 32 * H * 2);      // image splice rows
        // rows 33-128 decode on the row-block family, which exists for the Vicuna-7B widths only (api_dispatch.hip blk64_ok): say so HERE, not at the
        // first prefill of a context that already holds a 128-row KV cache
        if (B > 32 && !blk64_ok(c, B))
            return fail(c, -1, "rdx_finalize_weights: max_batch %d > 32 needs the Vicuna-7B widths (hidden 4096, inter 11008: the row-block decode family, model-dtype "
                               "or fp8 weights); this model (hidden %d, inter %d) holds at most 32 rows per context", B, H, I);
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

// Test / experiment switches of one context (read from the environment once, at rdx_create): "flash_min" = workgroups from which the batched
// prefill attention takes flash_prefill_k (0 = never, 1 = always; RDX_FLASH_MIN), "pconv" = the encoder on fragment-packed activations (RDX_PCONV),
// "xs16" = batch 3-16 decode on the one-row-tile family (RDX_XS16), "prompt_blk" = one prompt's K = 4096 projections on 32-row blocks (RDX_PBLK).
extern "C" int rdx_set_option(rdx_ctx* c, const char* name, int value) {
    if (!c || !name) return -1;
    if (!strcmp(name, "flash_min")) { c->flash_min = value; return 0; }
    if (!strcmp(name, "pconv")) { c->trunk_packed = value != 0; return 0; }
    if (!strcmp(name, "prompt_blk")) { c->prompt_blk = value != 0; return 0; }
    if (!strcmp(name, "xs16")) {          // batch 3-16 decode family (xs16.hip) on / off; a captured step graph of the other family is dropped
        c->xs16 = value != 0;
        if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; c->gkey = GraphKey(); }
        return 0;
    }
    return fail(c, -1, "rdx_set_option: unknown option '%s'", name);
}

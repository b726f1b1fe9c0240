// librdx_hooks.so: kernel-test, trace and microbenchmark hooks (include/rdx_hooks.h) -- NOT part of the product library. Built from this one
// file and linked against librdx.so, whose context and launchers it drives; radialog_amd/_lib.py loads it only under RDX_DEBUG_HOOKS=1 (the
// tests' conftest and the tools/ scripts set it). The product entry points, rdx_time included, are in librdx.so (include/rdx.h).
#include "rdx_ctx.h"
#include "../../include/rdx_hooks.h"

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

// Test hook: the fp8 path's activation quantisers on caller data. mode 0 = quant_rows_k (rows of X [M][K] -> e4m3 bytes [M][K] + scales [M][groups]),
// mode 1 = RMSNorm -> e4m3 (rmsnorm4096_k / rmsnorm_k <T, 5>: bytes [M][K] + one scale per row; norm_w [K] in the model dtype, groups ignored).
extern "C" int rdx_quant_test(rdx_ctx* c, const void* X, int M, int K, int groups, int mode, const void* norm_w, float eps, void* out8, float* scales) {
    if (!c || !X || !out8 || !scales || M <= 0 || K <= 0 || K % 128) return fail(c, -1, "rdx_quant_test: bad arguments (K %% 128 == 0)");
    if (mode == 0 && (groups < 1 || groups > 4)) return fail(c, -1, "rdx_quant_test: 1..4 K groups");
    if (mode == 1 && !norm_w) return fail(c, -1, "rdx_quant_test: mode 1 needs the norm weight");
    HIPCHK(c, hipSetDevice(c->device));
    if (mode == 0) launch_quant_rows(c->cfg.dtype, X, K, out8, scales, M, K, groups, c->stream);
    else launch_rmsnorm_fp8(c->cfg.dtype, X, norm_w, out8, scales, M, K, eps, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
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

// One NHWC convolution on caller data through (path 0) the production dispatch of the row-major encoder path (conv_gemm) or (path 1) the
// fragment-packed kernel family pconv_k (pack_rows -> pconv -> unpack_rows; path 2: pconv writing row-major itself). X [B][H][H][Cin] and resid /
// out [B][Ho][Ho][Cout] model dtype, W [Cout][ksize * ksize * Cin] fp32 in the (kh, kw, c) K order, bias fp32. ms_host (nullable) = ms per launch
// of the convolution kernel alone over `iters` launches (layout conversions excluded).
extern "C" int rdx_conv_test(rdx_ctx* c, const void* X, const float* W, const float* bias, const void* resid, void* out, int B, int H, int Cin,
                             int Cout, int ksize, int stride, int epi, int path, int iters, float* ms_host) {
    if (!c || !X || !W || !out || B <= 0) return fail(c, -1, "rdx_conv_test: bad arguments");
    if (!(ksize == 1 || ksize == 3)) return fail(c, -1, "rdx_conv_test: ksize 1 or 3");
    HIPCHK(c, hipSetDevice(c->device));
    const int K = ksize * ksize * Cin, Ho = (H + 2 * (ksize / 2) - ksize) / stride + 1;
    const int M = B * Ho * Ho, Min = B * H * H, dt = c->cfg.dtype;
    if (K % 32 || Cout % 16) return fail(c, -1, "rdx_conv_test: need K %% 32 == 0 and Cout %% 16 == 0");
    const bool need_res = epi == EPI_RESID || epi == EPI_RESID_RELU;
    if (need_res && !resid) return fail(c, -1, "rdx_conv_test: epilogue needs a residual");
    const int mt_in = (Min + 15) / 16, mt_out = (M + 15) / 16;
    const size_t wb = (size_t)Cout * K * 2, xpb = (size_t)mt_in * 16 * Cin * 2, opb = (size_t)mt_out * 16 * Cout * 2;
    char* buf = nullptr;
    HIPCHK(c, hipMalloc((void**)&buf, wb + xpb + 2 * opb + 256));
    GemmW w; w.N = Cout; w.K = K; w.Npad = Cout; w.w = buf;
    launch_pack_weight(dt, W, buf, Cout, K, Cout, nullptr, c->stream);
    char *xp = buf + wb, *rp = xp + xpb, *op = rp + opb;
    struct WsScope { rdx_ctx* c; ~WsScope() { c->ws_ok = false; } } ws_scope{c};
    c->ws_ok = true;
    PConvArgs pa;
    memset(&pa, 0, sizeof(pa));
    if (path) {
        launch_pack_rows(dt, X, Cin, xp, Min, Cin, c->stream);
        if (need_res) launch_pack_rows(dt, resid, Cout, rp, M, Cout, c->stream);
        pa.X = xp; pa.W = buf; pa.bias = bias; pa.resid = need_res ? rp : nullptr; pa.out = path == 2 ? out : (void*)op; pa.zero16 = c->zero16;
        pa.Hin = H; pa.Win = H; pa.Cin = Cin; pa.Hout = Ho; pa.Wout = Ho; pa.N = Cout; pa.M = M; pa.mt_in = mt_in; pa.mt_out = mt_out; pa.ldo = Cout;
        pa.no_ksplit = c->pconv_noks;
        if (const char* e = getenv("RDX_PCONV_TILE")) {            // "MxN" or "MxNkS": forced register tile (tools/pconv_check.py)
            if (e[0] >= '1' && e[0] <= '8' && e[1] == 'x' && (e[2] == '2' || e[2] == '4')) {
                pa.tile_m = e[0] - '0'; pa.tile_n = e[2] - '0';
                if (e[3] == 'k' && (e[4] == '4' || e[4] == '8')) pa.tile_k = e[4] - '0';
            }
        }
        if (!c->zero16 || !pconv_supported(pa, ksize * ksize, stride, epi)) { hipFree(buf); return fail(c, -1, "rdx_conv_test: shape not supported by pconv"); }
    }
    bool refused = false;
    auto once = [&]() {
        if (path) { if (!launch_pconv(dt, pa, ksize * ksize, stride, epi, path == 2, c->stream)) refused = true; }
        else conv_gemm(c, X, w, bias, need_res ? resid : nullptr, out, B, H, H, Cin, ksize, ksize, stride, ksize / 2, Ho, Ho, epi);
    };
    once();
    if (refused) { hipStreamSynchronize(c->stream); hipFree(buf); return fail(c, -1, "rdx_conv_test: pconv refused the tiling (1 x 1 stride-1 with different input / output row tilings)"); }
    if (ms_host && iters > 0) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, c->stream);
        for (int i = 0; i < iters; ++i) once();
        hipEventRecord(e1, c->stream);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipEventDestroy(e0); hipEventDestroy(e1);
        *ms_host = ms / (float)iters;
    }
    if (path == 1) launch_unpack_rows(dt, op, out, Cout, M, Cout, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(buf);
    HIPCHK(c, e);
    HIPCHK(c, hipGetLastError());
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

// fp8 weights at batch >= 3 multiply fp8 x fp8 (xstat32_k<W8, A8>): [RMSNorm +] e4m3 quantisation of the rows into the 32-row fragment block,
// like skinny_prenorm does in the decode step; `tmp` (caller frees) holds the block and its 32 scales
static int fp8_block_for_test(rdx_ctx* c, GemmArgs& a, int epi, char** tmp) {
    *tmp = nullptr;
    if (!(a.W8 && a.wscale) || a.M < xs_min_rows()) return 0;
    GemmArgs t = a; t.norm_w = nullptr;
    if (!xstat32_supported(t, epi)) return 0;
    HIPCHK(c, hipMalloc((void**)tmp, (size_t)32 * a.K + 32 * sizeof(float)));
    float* xs = (float*)(*tmp + (size_t)32 * a.K);
    launch_rmsnorm_packed32_fp8(c->cfg.dtype, const_cast<void*>(a.X), a.norm_w, *tmp, xs, a.M, a.K, a.eps, nullptr, 0, c->stream);
    a.X = *tmp; a.ldx = a.K; a.xpacked = 4; a.xscale = xs; a.xgroups = 1; a.norm_w = nullptr;
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
    HIPCHK(c, hipMalloc(&wp, (size_t)N * K * 2 + ((force == 4 || (force >= 9 && force <= 11)) ? (size_t)N * K + (size_t)N * 4 : 0)));
    w.w = wp;
    if (force >= 9 && force <= 11) {
        // the fp8 path's prefill GEMM (gemm8.hip): e4m3 weights x e4m3 activations with 1 / 2 / 4 K groups; RMSNorm -> fp8 (one group) or quant_rows_k
        const int G = force == 9 ? 1 : (force == 10 ? 2 : 4);
        if (K % 64 || (norm_w && G != 1)) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: gemm8 needs K %% 64 == 0 (and one K group behind an RMSNorm)"); }
        w.w = nullptr; w.w8 = (char*)wp + (size_t)N * K * 2; w.scale = (float*)((char*)wp + (size_t)N * K * 3);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, nullptr, N, K, N, c->stream);
        char* tmp = nullptr;
        if (hipMalloc((void**)&tmp, (size_t)M * K + (size_t)M * 4 * sizeof(float)) != hipSuccess) { hipFree(wp); return fail(c, -2, "rdx_gemm_test: out of device memory"); }
        float* xs = (float*)(tmp + (size_t)M * K);
        if (norm_w) launch_rmsnorm_fp8(c->cfg.dtype, X, norm_w, tmp, xs, M, K, eps, c->stream);
        else launch_quant_rows(c->cfg.dtype, X, K, tmp, xs, M, K, G, c->stream);
        GemmArgs a = gargs(tmp, K, w, nullptr, out, epi == EPI_SILU_MUL ? N / 2 : N, M);
        a.resid = resid; a.ldr = N; a.xscale = xs; a.xgroups = G;
        int rc = 0;
        if (!gemm8_supported(a, epi)) rc = fail(c, -1, "rdx_gemm_test: shape / epilogue not supported by gemm8");
        else launch_gemm8(c->cfg.dtype, a, epi, c->stream);
        hipError_t se = hipStreamSynchronize(c->stream);
        hipFree(wp); hipFree(tmp);
        if (rc) return rc;
        HIPCHK(c, se);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    if (force == 4) {                     // fp8 weights as the engine holds them: e4m3 bytes + scales only (no model-dtype copy)
        if (K % 64) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: fp8 needs K %% 64 == 0"); }
        w.w = nullptr;
        w.w8 = (char*)wp + (size_t)N * K * 2;
        w.scale = (float*)((char*)wp + (size_t)N * K * 3);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, nullptr, N, K, N, c->stream);
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
        w.w = nullptr; w.w8 = (char*)wp + (size_t)N * K * 2; w.scale = (float*)((char*)wp + (size_t)N * K * 3);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, nullptr, N, K, N, c->stream);
        a = gargs(X, K, w, bias, out, N, M);
        a.resid = resid; a.ldr = N; a.eps = eps;
        split8 = true; force = 5;
    }
    if (force == 5) {       // K-split slab path: pack X -> xsplit32_k -> slab combine (+ residual) at the launch boundary; out = resid + T(X W^T)
        if (epi != EPI_RESID || !resid || M <= 16 || M > 32) { hipFree(wp); return fail(c, -1, "rdx_gemm_test: force 5 needs epi 3 and 16 < M <= 32"); }
        char* tmp = nullptr;
        const size_t xb = (size_t)32 * K * 2, sb = (size_t)4 * 32 * N * 4;
        if (hipMalloc((void**)&tmp, 2 * xb + sb) != hipSuccess) { hipFree(wp); return fail(c, -2, "rdx_gemm_test: out of device memory"); }
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
        char* blk = nullptr;
        int brc = fp8_block_for_test(c, a, epi, &blk);
        if (brc) { hipFree(wp); return brc; }
        if (a.norm_w && !skinny_fits_lds(M, K)) {
            HIPCHK(c, hipMalloc(&xn, (size_t)32 * K * 2));
            c->dxn = xn;
        }
        skinny(c, a, epi == EPI_RESID_RELU ? EPI_RESID : epi);
        c->dxn = keep;
        hipError_t se = hipStreamSynchronize(c->stream);
        if (blk) hipFree(blk);
        if (int urc = take_unsupported(c)) { hipFree(wp); if (xn) hipFree(xn); return urc; }
        HIPCHK(c, se);
    } else {
        if (a.norm_w) {
            HIPCHK(c, hipMalloc(&xn, (size_t)M * K * 2));
            launch_rmsnorm(c->cfg.dtype, X, norm_w, xn, M, K, eps, c->stream);
            a.X = xn; a.norm_w = nullptr;
        }
        ConvGeom cg;
        memset(&cg, 0, sizeof(cg));
        if (force == 3 || (force == 0 && gemm_dma_supported(a))) {
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
        w.w = nullptr; w.w8 = wp + wb; w.scale = (float*)(wp + wb + (size_t)N * K);
        launch_pack_weight_fp8(c->cfg.dtype, W, w.w8, w.scale, nullptr, N, K, N, c->stream);
    } else {
        launch_pack_weight(c->cfg.dtype, W, wp, N, K, N, nullptr, c->stream);
    }
    GemmArgs a = gargs(X, K, w, nullptr, out_logits, N, M);
    a.n_valid = n_valid;
    a.part_val = (float*)(wp + wb + qb); a.part_idx = (int*)(wp + wb + qb + pb);
    char* blk = nullptr;
    int brc = fp8_block_for_test(c, a, EPI_LOGITS, &blk);
    if (brc) { hipFree(wp); return brc; }
    skinny(c, a, EPI_LOGITS);
    std::vector<float> pv((size_t)M * nt);
    std::vector<int> pi((size_t)M * nt);
    hipError_t se = hipStreamSynchronize(c->stream);
    if (blk) hipFree(blk);
    if (int urc = take_unsupported(c)) { hipFree(wp); return urc; }
    HIPCHK(c, se);
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


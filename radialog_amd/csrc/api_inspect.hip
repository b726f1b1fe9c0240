// librdx C ABI, part 5: introspection and in-situ timing of the hot-path units (rdx_kv_read / rdx_hidden_read: the parity tests' view of the
// decoder state; rdx_time: HIP-event timing on the context's stream -- bench.py's live `roofline` figures). Part of the product library.
// (The kernel-test / trace / microbenchmark hooks live in api_debug.hip -> librdx_hooks.so, loaded only under RDX_DEBUG_HOOKS=1.)
#include "rdx_ctx.h"

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
    if (B > 32 && what >= 1 && what <= 5) return fail(c, -1, "rdx_time: the per-projection units time the <= 32-row kernel families; at %d rows use unit 0 (step) or 6 (attention)", B);
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    int launches = 0;
    if (what == 7) {
        // the chained down(l) -> QKV(l+1) launch (decode_chain_k, batch <= 2), IN SITU: `iters` eager decode steps with an event pair
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
        GemmArgs pn;
        auto pre = [&](GemmArgs a, int epi) {
            if (xpk < 0) { pn = skinny_prenorm(c, a, epi); xpk = pn.norm_w ? 0 : (pn.X == c->dxn ? 1 + pn.xpacked : 0); }
            if (xpk > 0) { a.X = c->dxn; a.ldx = a.K; a.norm_w = nullptr; a.xpacked = xpk - 1; a.xscale = pn.xscale; a.xgroups = pn.xgroups; }
            return a;
        };
        // batch 3-16 decodes on xs16.hip (norm-prologue xstat16_k, un-split xrow16_k): time THOSE kernels, not the 32-row family the step never launches there
        const bool x16 = what >= 1 && what <= 5 && xs16_ok(c, B);
        if (x16) {
            HIPCHK(c, hipEventRecord(e0, c->stream));
            for (int i = 0; i < iters; ++i) {
                for (int l = 0; l < (what == 5 ? 1 : f.layers); ++l) {
                    const LlamaLayer& L = c->ll[same_layer ? 0 : l];
                    if (what == 1) { GemmArgs a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps; a.out_packed = 1; xs16_proj(c, a, EPI_SILU_MUL); }
                    else if (what == 2) { GemmArgs a = gargs(c->dx, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; a.norm_w = L.attn_norm; a.eps = f.rms_eps; xs16_proj(c, a, EPI_NONE); }
                    else if (what == 5) { GemmArgs a = gargs(c->dx, H, c->lm_head, nullptr, nullptr, f.vocab, B); a.N = c->lm_head.Npad; a.n_valid = f.vocab; a.norm_w = c->final_norm; a.eps = f.rms_eps;
                                          a.part_val = c->part_val; a.part_idx = c->part_idx; xs16_proj(c, a, EPI_LOGITS); }
                    else {      // 3: o_proj, 4: down_proj -- xrow16_k with the residual epilogue, written to a scratch buffer so that repeated launches do not grow dx
                        const GemmW& W = what == 3 ? L.wo : L.wdown;
                        GemmArgs a = gargs(what == 3 ? c->datt : c->dgu, W.K, W, nullptr, c->dqkv, H, B); a.resid = c->dx; a.ldr = H; a.xpacked = 1;
                        launch_xrow16(f.dtype, a, c->stream);
                    }
                    ++launches;
                }
            }
            HIPCHK(c, hipEventRecord(e1, c->stream));
            HIPCHK(c, hipEventSynchronize(e1));
            float ms16 = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms16, e0, e1));
            hipEventDestroy(e0); hipEventDestroy(e1);
            *ms_host = ms16 / (float)launches;
            return take_unsupported(c);
        }
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

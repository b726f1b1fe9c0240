// librdx C ABI, part 3: the Llama decoder -- prompt prefill, the hipGraph-captured greedy decode step, multi-turn append, beam search.
#include "rdx_ctx.h"

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
    dfree(c, c->px); dfree(c, c->pxn); dfree(c, c->pqkv); dfree(c, c->pq); dfree(c, c->patt); dfree(c, c->pgu); dfree(c, c->pxq); dfree(c, c->pxs); dfree(c, c->pslab);
    if (fp8_weights(c->ll[0].wqkv)) {       // e4m3 activations of the fp8 x fp8 prefill GEMMs (gemm8.hip) and their per-(row, K group) scales
        ALLOC(c, c->pxq, rows * (size_t)std::max(f.hidden, f.inter));
        ALLOC(c, c->pxs, rows * 4 * sizeof(float));
    }
    ALLOC(c, c->px, rows * f.hidden * 2); ALLOC(c, c->pxn, rows * f.hidden * 2);
    ALLOC(c, c->pslab, (size_t)4 * std::min<size_t>(rows, 192) * f.hidden * sizeof(float));
    ALLOC(c, c->pqkv, rows * c->ld.qkv_ld * 2); ALLOC(c, c->pq, rows * f.hidden * 2);
    ALLOC(c, c->patt, rows * f.hidden * 2); ALLOC(c, c->pgu, rows * f.inter * 2);
    c->prefill_rows = rows;
    return 0;
}

static void lm_head_and_greedy(rdx_ctx* c, const void* x, int B, void* logits, const int* out_step, long step_stride,
                               int advance) {
    const rdx_config& f = c->cfg;
    GemmArgs a = gargs(x, f.hidden, c->lm_head, nullptr, logits, f.vocab, B);
    a.N = c->lm_head.Npad; a.n_valid = f.vocab;
    a.norm_w = c->final_norm; a.eps = f.rms_eps;
    a.part_val = c->part_val; a.part_idx = c->part_idx;
    a.out_step = out_step; a.out_step_stride = step_stride;
    if (B > 32) {
        // 33-128 rows: final RMSNorm into the fragment-packed row tiles, lm_head in two row blocks (xstat32_k<EPI_LOGITS, BLK>)
        if (!blk64_ok(c, B)) { c->unsupported = "more than 32 decoder rows need hidden 4096 / inter 11008 (the row-block family)"; return; }
        const int mtl = (B + 15) / 16;
        const int pend = (x == c->dx) ? c->pend_groups : 0;       // the last layer's K-split down_proj left its slabs (and the residual add) to this norm
        if (blk64_fp8(c)) {
            if (pend) c->pend_groups = 0;
            launch_rmsnorm_blk_fp8(f.dtype, const_cast<void*>(x), c->final_norm, c->dxn, c->dxs, B, mtl, f.rms_eps, pend ? c->kslab : nullptr, pend, c->stream);
            a.X = c->dxn; a.norm_w = nullptr; a.xpacked = 4; a.mtiles = mtl; a.xscale = c->dxs; a.xgroups = 1;
            launch_xstat_blk8(f.dtype, a, EPI_LOGITS, c->stream);
        }
        else if (pend) { c->pend_groups = 0; launch_rmsnorm_packed_slab(f.dtype, c->dx, c->final_norm, c->dxn, B, mtl, f.rms_eps, c->kslab, pend, c->stream); }
        else launch_rmsnorm_packed(f.dtype, x, c->final_norm, c->dxn, B, mtl, f.hidden, f.rms_eps, c->stream);
        if (!blk64_fp8(c)) {
            a.X = c->dxn; a.norm_w = nullptr; a.xpacked = 3; a.mtiles = mtl;
            launch_xstat_blk(f.dtype, a, EPI_LOGITS, c->stream);
        }
    }
    else if (x == c->dx && xs16_ok(c, B)) xs16_proj(c, a, EPI_LOGITS);        // batch 3-16 decode: the final RMSNorm is the kernel's prologue
    else skinny(c, a, EPI_LOGITS);
    launch_greedy_step(f.dtype, c->part_val, c->part_idx, c->n_vtiles, B, c->cur_eos, c->cur_pad, c->cur_max_new,
                       c->cur_tokens, c->d_unf, advance ? c->d_pos : nullptr, advance ? c->d_slot : nullptr, c->d_step,
                       c->embed, f.vocab, c->dx, f.hidden, c->d_pos, c->rope_cos, c->rope_sin, c->d_cur_rope,
                       (c->fuse_attn_oproj || c->chain_mlp) ? c->d_ctr : nullptr,
                       f.layers * 256 + (c->chain_mlp ? (int)chain_ctr_ints(f.layers) : 0), c->stream);
}

// keep == 0: a fresh prompt. keep > 0: `T` further prompt tokens behind the first `keep` cache slots of the previous call(s)
// (the shared prefix of a multi-turn conversation is not recomputed; no image splice in the continuation).
int prefill_impl(rdx_ctx* c, const int32_t* ids, const int32_t* mask, int B, int T, const float* qformer_embs, int keep,
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

    const bool fp8 = fp8_weights(c->ll[0].wqkv);
    if (B > 32 && !blk64_ok(c, B)) return fail(c, -8, "rdx_prefill: more than 32 rows per context need the Vicuna-7B widths (hidden 4096, inter 11008: the row-block decode family)");
    if (fp8) {
        // fp8 weights (BASELINE configs[4]): every projection of the prompt is an fp8 x fp8 MFMA GEMM (gemm8.hip) over e4m3 activations with one
        // scale per row and K group -- 1 group behind an RMSNorm (quantised in its epilogue), 2 for o_proj, 4 for down_proj (quant_rows_k on the
        // attention / SwiGLU output); LoRA-B, RoPE, attention, SwiGLU and the residual adds stay in the model dtype
        for (int l = 0; l < f.layers; ++l) {
            const LlamaLayer& L = c->ll[l];
            void* kc = kv_ptr(c, c->kcache, l);
            void* vc = kv_ptr(c, c->vcache, l);
            auto g8 = [&](const GemmW& W, int K, int groups, void* out, int ldo, const void* resid, int epi) {
                GemmArgs a = gargs(c->pxq, K, W, nullptr, out, ldo, (int)M);
                a.N = W.Npad; a.xscale = c->pxs; a.xgroups = groups; a.resid = resid; a.ldr = H;
                if (!gemm8_supported(a, epi)) { c->unsupported = "fp8 weights: prefill projection shape not supported by gemm8 (N % 16, K % 64)"; return; }
                launch_gemm8(dt, a, epi, s);
            };
            launch_rmsnorm_fp8(dt, c->px, L.attn_norm, c->pxq, c->pxs, (int)M, H, f.rms_eps, s);
            g8(L.wqkv, H, 1, c->pqkv, c->ld.qkv_ld, nullptr, EPI_NONE);
            launch_rope_kv_prefill(dt, c->ld, c->pqkv, L.lora_bq, L.lora_bv, c->rope_cos, c->rope_sin, c->d_pos_ids, c->pq, kc, vc, B, T, keep, s);
            AttnArgs at;
            memset(&at, 0, sizeof(at));
            at.Q = c->pq; at.q_bs = (long)T * H; at.q_ts = H; at.q_hs = 128;
            at.K = kc; at.V = vc; at.k_bs = at.v_bs = (long)f.heads * f.max_len * 128; at.k_ts = at.v_ts = 128; at.k_hs = at.v_hs = (long)f.max_len * 128;
            at.O = c->patt; at.o_bs = (long)T * H; at.o_ts = H; at.o_hs = 128;
            at.B = B; at.H = f.heads; at.Tq = T; at.Tk = keep + T; at.causal = 1; at.k_perm = c->ld.k_perm; at.key_mask = c->key_mask; at.km_bs = f.max_len;
            at.flash_min = c->flash_min;
            launch_attention(dt, 128, at, s);
            launch_quant_rows(dt, c->patt, H, c->pxq, c->pxs, (int)M, H, 2, s);
            g8(L.wo, H, 2, c->px, H, c->px, EPI_RESID);
            launch_rmsnorm_fp8(dt, c->px, L.mlp_norm, c->pxq, c->pxs, (int)M, H, f.rms_eps, s);
            g8(L.wgu, H, 1, c->pgu, f.inter, nullptr, EPI_SILU_MUL);
            launch_quant_rows(dt, c->pgu, f.inter, c->pxq, c->pxs, (int)M, f.inter, 4, s);
            g8(L.wdown, f.inter, 4, c->px, H, c->px, EPI_RESID);
        }
    } else {
    // few rows (one or two prompts): the projections are weight-stream bound -> weight-stationary kernels over fragment-packed
    // activations (wstat.hip); the producers (RMSNorm, attention, the SwiGLU epilogue) write that order directly
    const int mtl = (int)((M + 15) / 16);
    constexpr int ws_maxm = 384;
    // measured (tools/prefill_only.py, 32 layers): wstat's time grows with the row tiles of 16, the 128-row tile GEMMs' with the row tiles of 128 --
    // M = 64: 4.84 vs 5.34 ms, 100: 6.06 / 6.26, 160: 7.30 / 7.64, 320: 11.43 / 11.92, but 250: 9.56 / 8.86. Take wstat when the 128-row
    // tiling would pad by 24 rows or more.
    constexpr int ws_minpad = 24;
    bool ws = (int)M > 32 && (int)M <= ws_maxm && (int)((M + 127) / 128 * 128 - M) >= ws_minpad;
    if (ws) {
        GemmArgs p = gargs(c->pxn, H, c->ll[0].wqkv, nullptr, c->pqkv, c->ld.qkv_ld, (int)M); p.xpacked = 3; p.mtiles = mtl;
        GemmArgs d = gargs(c->pgu, f.inter, c->ll[0].wdown, nullptr, c->px, H, (int)M); d.xpacked = 3; d.mtiles = mtl;
        ws = wstat_supported(p, EPI_NONE) && wstat_supported(d, EPI_RESID);
    }
    auto prompt_gemm = [&](GemmArgs a, int epi, bool packed_out) {
        if (!ws) { run_gemm(c, a, epi); return; }
        a.xpacked = 3; a.mtiles = mtl; a.out_packed = packed_out ? 3 : 0;
        // round 5: up to 192 rows the K = 4096 projections run activation-stationary in row blocks of 32 whose workgroups share an XCD's L2
        // (xstat32_k<.., BLK>: the weights cross each CU once per row block instead of the activations once per 32 columns); down_proj
        // (K = 11008) and longer prompts keep the weight-stationary kernel
        if (c->prompt_blk && xstat_blk_supported(a, epi)) { launch_xstat_blk(dt, a, epi, s); return; }
        launch_wstat(dt, a, epi, s);
    };
    // round 5: down_proj of a prompt of <= 128 rows (<= 4 row blocks: 160 rows measured 55.9 us against wstat_k 49.5; 64 rows 3.96 -> 3.69 ms per prefill) K-split over 4 workgroups per tile into fp32 slabs (xsplit32_k<.., BLK>), combined (+ residual) by the next
    // layer's RMSNorm -- after the last layer by one more norm launch whose packed output nobody reads
    int pend = 0;
    for (int l = 0; l < f.layers; ++l) {
        const LlamaLayer& L = c->ll[l];
        void* kc = kv_ptr(c, c->kcache, l);
        void* vc = kv_ptr(c, c->vcache, l);
        if (ws && pend) { launch_rmsnorm_packed_slab(dt, c->px, L.attn_norm, c->pxn, (int)M, mtl, f.rms_eps, c->pslab, pend, s); pend = 0; }
        else if (ws) launch_rmsnorm_packed(dt, c->px, L.attn_norm, c->pxn, (int)M, mtl, H, f.rms_eps, s);
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
        at.flash_min = c->flash_min;
        launch_attention(dt, 128, at, s);
        { GemmArgs a = gargs(c->patt, H, L.wo, nullptr, c->px, H, (int)M); a.resid = c->px; a.ldr = H; prompt_gemm(a, EPI_RESID, false); }
        if (ws) launch_rmsnorm_packed(dt, c->px, L.mlp_norm, c->pxn, (int)M, mtl, H, f.rms_eps, s);
        else launch_rmsnorm(dt, c->px, L.mlp_norm, c->pxn, (int)M, H, f.rms_eps, s);
        { GemmArgs a = gargs(c->pxn, H, L.wgu, nullptr, c->pgu, f.inter, (int)M); prompt_gemm(a, EPI_SILU_MUL, true); }
        { GemmArgs a = gargs(c->pgu, f.inter, L.wdown, nullptr, c->px, H, (int)M); a.resid = c->px; a.ldr = H;
          GemmArgs b = a; b.xpacked = 3; b.mtiles = mtl;
          if (ws && c->prompt_blk && M <= 128 && xsplit_blk_supported(b)) { launch_xsplit_blk(dt, b, c->pslab, s); pend = 4; }
          else prompt_gemm(a, EPI_RESID, false); }
    }
    if (pend) launch_rmsnorm_packed_slab(dt, c->px, c->ll[0].attn_norm, c->pxn, (int)M, mtl, f.rms_eps, c->pslab, pend, s);
    }
    launch_gather_last(dt, c->px, c->datt, B, T, H, s);      // datt doubles as the [B][H] last-position buffer
    lm_head_and_greedy(c, c->datt, B, logits, nullptr, 0, /*advance=*/0);
    HIPCHK(c, hipGetLastError());
    return take_unsupported(c);
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

bool decode_step_launch(rdx_ctx* c, void* logits, const int* out_step, long step_stride, std::vector<hipEvent_t>* evs) {
    const rdx_config& f = c->cfg;
    const int dt = f.dtype, H = f.hidden, B = c->cur_B;
    hipStream_t s = c->stream;
    // the hand-off counter shards of the fused launches are cleared by greedy_step_k at the end of the previous step
    // (and of the prefill): a memset node at the head of the step graph was observed to race with the first producers
    // batch <= 2: down(l) -> QKV(l+1) chained inside one launch by the fence-free hand-off, attention + o_proj in the fused launch
    const bool chain = c->chain_mlp && c->fuse_attn_oproj && chain_supported(c->ld, f.inter, B) &&
                       attn_oproj16_supported(c->ld, f.hidden, f.hidden, B);
    ChainArgs ca;
    if (chain) {
        memset(&ca, 0, sizeof(ca));
        ca.layers = c->d_clayers; ca.hidden = H; ca.inter = f.inter; ca.qkv_n = c->ll[0].wqkv.Npad; ca.qkv_ld = c->ld.qkv_ld; ca.B = B; ca.eps = f.rms_eps;
        ca.dx = c->dx; ca.dqkv = c->dqkv; ca.dgu = c->dgu; ca.ctr = c->d_cctr; ca.err = c->d_err; ca.naps = c->chain_naps;
        const LlamaLayer& L0 = c->ll[0];        // fp8 weights: the chained roles stream the e4m3 bytes too
        ca.w8 = (L0.wqkv.w8 && L0.wdown.w8 && f.hidden % 64 == 0 && f.inter % 64 == 0) ? 1 : 0;
    }
    if (B > 32) {
        // 33-128 rows: the row-block family (api_dispatch.hip blk64_ok): 7 launches per layer, no K-split slabs
        if (!blk64_ok(c, B)) { c->unsupported = "more than 32 decoder rows need hidden 4096 / inter 11008 (the row-block family)"; return false; }
        const int mtl = (B + 15) / 16;
        if (blk64_fp8(c)) {
            // fp8 x fp8: the 32-row fp8 kernels per row block (api_dispatch.hip blk64_ok); both residual projections K-split, their slabs + residual left to the next RMSNorm
            auto p8 = [&](GemmArgs a, int xp, int outp) { a.xpacked = xp; a.mtiles = mtl; a.out_packed = outp; a.xscale = c->dxs; a.xgroups = 1; return a; };
            for (int l = 0; l < f.layers; ++l) {
                const LlamaLayer& L = c->ll[l];
                { const int pend = c->pend_groups; c->pend_groups = 0;
                  launch_rmsnorm_blk_fp8(dt, c->dx, L.attn_norm, c->dxn, c->dxs, B, mtl, f.rms_eps, pend ? c->kslab : nullptr, pend, s); }
                { GemmArgs a = gargs(c->dxn, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; launch_xstat_blk8(dt, p8(a, 4, 0), EPI_NONE, s); }
                DecAttnArgs at;
                at.d = c->ld; at.qkv = c->dqkv; at.lbq = L.lora_bq; at.lbv = L.lora_bv; at.cos_t = c->rope_cos; at.sin_t = c->rope_sin;
                at.pos = c->d_pos; at.slot_b = c->d_slot; at.key_mask = c->key_mask; at.cur_rope = c->d_cur_rope;
                at.kcache = kv_ptr(c, c->kcache, l); at.vcache = kv_ptr(c, c->vcache, l); at.out = c->datt;
                at.out_packed = 2;                       // the 64-deep order, one 32-row block per 32 rows
                launch_decode_attention(dt, at, B, s);
                { GemmArgs a = gargs(c->datt, H, L.wo, nullptr, c->dx, H, B); launch_xsplit_blk8(dt, p8(a, 2, 0), c->kslab, s); }
                launch_rmsnorm_blk_fp8(dt, c->dx, L.mlp_norm, c->dxn, c->dxs, B, mtl, f.rms_eps, c->kslab, 2, s);
                { GemmArgs a = gargs(c->dxn, H, L.wgu, nullptr, c->dgu, f.inter, B); launch_xstat_blk8(dt, p8(a, 4, 2), EPI_SILU_MUL, s); }
                { GemmArgs a = gargs(c->dgu, f.inter, L.wdown, nullptr, c->dx, H, B); launch_xsplit_blk8(dt, p8(a, 2, 0), c->kslab, s); c->pend_groups = 4; }
            }
            lm_head_and_greedy(c, c->dx, B, logits, out_step, step_stride, /*advance=*/1);
            return false;
        }
        auto pk = [&](GemmArgs a, int outp) { a.xpacked = 3; a.mtiles = mtl; a.out_packed = outp; return a; };
        for (int l = 0; l < f.layers; ++l) {
            const LlamaLayer& L = c->ll[l];
            if (c->pend_groups) { launch_rmsnorm_packed_slab(dt, c->dx, L.attn_norm, c->dxn, B, mtl, f.rms_eps, c->kslab, c->pend_groups, s); c->pend_groups = 0; }
            else launch_rmsnorm_packed(dt, c->dx, L.attn_norm, c->dxn, B, mtl, H, f.rms_eps, s);
            { GemmArgs a = gargs(c->dxn, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; launch_xstat_blk(dt, pk(a, 0), EPI_NONE, s); }
            DecAttnArgs at;
            at.d = c->ld; at.qkv = c->dqkv; at.lbq = L.lora_bq; at.lbv = L.lora_bv; at.cos_t = c->rope_cos; at.sin_t = c->rope_sin;
            at.pos = c->d_pos; at.slot_b = c->d_slot; at.key_mask = c->key_mask; at.cur_rope = c->d_cur_rope;
            at.kcache = kv_ptr(c, c->kcache, l); at.vcache = kv_ptr(c, c->vcache, l); at.out = c->datt;
            at.out_packed = 1; at.out_mt = mtl;
            launch_decode_attention(dt, at, B, s);
            { GemmArgs a = gargs(c->datt, H, L.wo, nullptr, c->dx, H, B); a.resid = c->dx; a.ldr = H; launch_xstat_blk(dt, pk(a, 0), EPI_RESID, s); }
            launch_rmsnorm_packed(dt, c->dx, L.mlp_norm, c->dxn, B, mtl, H, f.rms_eps, s);
            { GemmArgs a = gargs(c->dxn, H, L.wgu, nullptr, c->dgu, f.inter, B); launch_xstat_blk(dt, pk(a, 3), EPI_SILU_MUL, s); }
            // down_proj: K-split over 4 workgroups per tile into fp32 slabs, combined (+ residual) by the next RMSNorm (xsplit32_k<.., BLK>: 21 us against
            // 29.5 us for the prompt's weight-stationary kernel at 64 rows); RDX_BLK_DOWN=0: wstat_k, the A/B leg
            { GemmArgs a = gargs(c->dgu, f.inter, L.wdown, nullptr, c->dx, H, B); a.resid = c->dx; a.ldr = H; a = pk(a, 0);
              static const bool split_down = !(getenv("RDX_BLK_DOWN") && atoi(getenv("RDX_BLK_DOWN")) == 0);
              if (split_down && xsplit_blk_supported(a)) { launch_xsplit_blk(dt, a, c->kslab, s); c->pend_groups = 4; }
              else launch_wstat(dt, a, EPI_RESID, s); }
        }
        lm_head_and_greedy(c, c->dx, B, logits, out_step, step_stride, /*advance=*/1);
        return false;
    }
    if (!chain && xs16_ok(c, B)) {
        // batch 3-16 (xs16.hip): five launches per layer -- QKV with the RMSNorm as its prologue, attention (output fragment-packed), o_proj with the
        // residual epilogue, gate/up with the RMSNorm prologue (SwiGLU output fragment-packed), down_proj with the residual epilogue
        for (int l = 0; l < f.layers; ++l) {
            const LlamaLayer& L = c->ll[l];
            { GemmArgs a = gargs(c->dx, H, L.wqkv, nullptr, c->dqkv, c->ld.qkv_ld, B); a.N = L.wqkv.Npad; a.norm_w = L.attn_norm; a.eps = f.rms_eps;
              xs16_proj(c, a, EPI_NONE); }
            DecAttnArgs at;
            at.d = c->ld; at.qkv = c->dqkv; at.lbq = L.lora_bq; at.lbv = L.lora_bv; at.cos_t = c->rope_cos; at.sin_t = c->rope_sin;
            at.pos = c->d_pos; at.slot_b = c->d_slot; at.key_mask = c->key_mask; at.cur_rope = c->d_cur_rope;
            at.kcache = kv_ptr(c, c->kcache, l); at.vcache = kv_ptr(c, c->vcache, l); at.out = c->datt;
            at.out_packed = 1;
            launch_decode_attention(dt, at, B, s);
            xs16_row(c, c->datt, L.wo, B);
            { GemmArgs a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps; a.out_packed = 1;
              xs16_proj(c, a, EPI_SILU_MUL); }
            xs16_row(c, c->dgu, L.wdown, B);
        }
        lm_head_and_greedy(c, c->dx, B, logits, out_step, step_stride, /*advance=*/1);
        return false;
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
        if (c->fuse_attn_oproj && attn_oproj16_supported(c->ld, L.wo.N, L.wo.K, B)) {
            launch_attn_oproj16(dt, at, ao, B, c->d_ctr + (size_t)l * 256, c->d_err, s);
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
        if (chain) {          // gate/up stand-alone, then down(l) -> QKV(l+1) chained
            { GemmArgs a = gargs(c->dx, H, L.wgu, nullptr, c->dgu, f.inter, B); a.norm_w = L.mlp_norm; a.eps = f.rms_eps; skinny(c, a, EPI_SILU_MUL); }
            if (evs) { hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s); evs->push_back(e); }
            ca.layer = l;
            launch_decode_chain(dt, ca, l + 1 < f.layers, s);
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
    return chain;
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
    return take_unsupported(c);
}

// The caller drives the loop and supplies the token itself (what LlamaForCausalLM.forward(input_ids = [B, 1], past_key_values = ...) is
// in the reference, modeling_llama_imgemb.py:705-793 behind prepare_inputs_for_generation :795-836): the embedding rows of `ids` replace
// the rows of the token the previous step selected, then the ordinary decode step runs.
extern "C" int rdx_decode_step_ids(rdx_ctx* c, const int32_t* ids, void* logits) {
    if (!c) return -1;
    if (!ids) return fail(c, -1, "rdx_decode_step_ids: null ids");
    if (!c->finalized || c->cur_B <= 0) return fail(c, -1, "rdx_decode_step_ids: no prefill has run");
    if (c->cur_steps >= c->cur_max_new)
        return fail(c, -1, "rdx_decode_step_ids: all %d tokens of this prompt (max_new) have been generated; run a new prefill", c->cur_max_new);
    if (c->cur_T + c->cur_steps > c->cfg.max_len || c->cur_T + c->cur_steps > c->cfg.max_pos)
        return fail(c, -1, "rdx_decode_step_ids: KV cache full (%d prompt + %d generated slots of %d)", c->cur_T, c->cur_steps, c->cfg.max_len);
    HIPCHK(c, hipSetDevice(c->device));
    launch_embed_rows(c->cfg.dtype, ids, c->embed, c->cfg.vocab, c->dx, c->cur_B, c->cfg.hidden, c->stream);
    decode_step_launch(c, logits, nullptr, 0);
    ++c->cur_steps;
    HIPCHK(c, hipGetLastError());
    return take_unsupported(c);
}

int build_graph(rdx_ctx* c, void* scores, bool fixed) {
    const rdx_config& f = c->cfg;
    GraphKey k;
    k.B = c->cur_B; k.max_new = c->cur_max_new; k.eos = c->cur_eos; k.pad = c->cur_pad; k.tokens = c->cur_tokens; k.scores = scores; k.fixed = fixed;
    if (c->graph && k == c->gkey) return 0;
    if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    hipGraph_t g = nullptr;
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    decode_step_launch(c, scores, (scores && !fixed) ? c->d_step : nullptr, (long)c->cur_B * f.vocab);
    HIPCHK(c, hipStreamEndCapture(c->stream, &g));
    if (int urc = take_unsupported(c)) { hipGraphDestroy(g); return urc; }
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
        // EOS poll: a 4-byte-per-row copy + stream sync every 4th step (round 4: every 16th -- up to 15 wasted 3.9-ms steps per finished
        // batch); the sync costs one launch-queue drain (~20 us) per 4 steps of 2.6-3.9 ms. RDX_EOS_POLL overrides (tools/eos_time.py)
        static const int check_every = [] { const char* e = getenv("RDX_EOS_POLL"); const int v = e ? atoi(e) : 4; return v > 0 ? v : 4; }();
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
        ALLOC(c, c->bm_tok, (size_t)R * 4); ALLOC(c, c->bm_src, (size_t)R * 2 * 4); ALLOC(c, c->bm_out, (size_t)R * N * 4);       // bm_src: src[R] | start[R]
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
    std::vector<int> ci((size_t)groups * K2), next_tok(rows), src(2 * (size_t)rows);       // src[rows] | first differing cache position[rows]
    std::vector<std::vector<int>> hist(rows), nh(rows);
    std::vector<BeamHyps> hyps(groups);
    for (BeamHyps& h : hyps) { h.num_beams = num_beams; h.length_penalty = length_penalty; h.early_stopping = early_stopping; }
    std::vector<char> done(groups, 0);
    const bool full_copy = getenv("RDX_BEAM_FULLCOPY") && atoi(getenv("RDX_BEAM_FULLCOPY"));     // tests: move every generated position (A/B leg)
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
        // _reorder_cache moves only what differs: row r's cache holds the positions of hist[r], its new parent's those of hist[src[r]];
        // both are token paths from the same prompt, so the common prefix is already in place (same tokens, same kernels, same launch)
        for (int r = 0; r < rows; ++r) {
            const std::vector<int>& mine = hist[r];
            const std::vector<int>& par = hist[src[r]];
            size_t cpre = 0;
            while (cpre < mine.size() && cpre < par.size() && mine[cpre] == par[cpre]) ++cpre;
            src[rows + r] = full_copy ? 0 : (T + (int)cpre) / 16 * 16;
        }
        for (int r = 0; r < rows; ++r) { nh[r] = hist[src[r]]; nh[r].push_back(next_tok[r]); }
        hist.swap(nh);
        ++cur_len;
        bool all_done = true;
        for (int g = 0; g < groups; ++g) all_done = all_done && done[g];
        if (all_done || cur_len >= T + max_new) break;
        // next forward: _reorder_cache (generated slots only -- the beams of a group share their prompt), chosen tokens in
        HIPCHK(c, hipMemcpyAsync(c->bm_src, src.data(), 2 * rows * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->bm_tok, next_tok.data(), rows * sizeof(int), hipMemcpyHostToDevice, s));
        bool identity = true;
        for (int r = 0; r < rows; ++r) identity = identity && src[r] == r;
        if (step > 0 && !identity)
            launch_kv_beam_reorder(c->kcache, c->vcache, c->bm_scratch, c->bm_src, c->bm_src + rows, rows, f.heads, f.layers, f.max_len, c->kv_layer_elems * 2,
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

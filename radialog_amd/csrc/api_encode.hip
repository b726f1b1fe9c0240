// librdx C ABI, part 2: image encode -- BioViL-T ResNet-50 trunk -> projector -> scramble + ln_vision -> Q-Former query path
// (rdx_encode_image / rdx_encode_image2 / rdx_classify_findings).
#include "rdx_ctx.h"

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
    mx += (size_t)16 * std::max(std::max(f.v_stem, f.v_planes[3] * 4), f.v_proj) * 2;       // fragment-packed tensors round their rows up to whole 16-row tiles
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
    // Round 4: the trunk runs on FRAGMENT-PACKED activations (pconv.hip: both MFMA operands straight from L2 into registers, no LDS, no
    // barriers; packed [C / 32][pixels / 16][64][8]) whenever every convolution of it has a packed kernel -- the stem writes the packed
    // layout, the last Bottleneck writes row-major NHWC for the GEMMs behind it. RDX_PCONV=0 keeps the row-major kernels of rounds 1-3
    // (the A/B leg of tests/test_gpu_fullsize.py).
    bool packed = c->zero16 && c->trunk_packed && stem_pool_supported(f.v_stem) && f.v_stem % 32 == 0;
    if (packed) {
        int Ch = f.v_stem;
        for (const VBlock& vb : c->vb) {
            PConvArgs t; memset(&t, 0, sizeof(t));
            t.Cin = Ch; t.N = vb.planes; packed = packed && pconv_supported(t, 1, 1, EPI_RELU);
            t.Cin = vb.planes; t.N = vb.planes; packed = packed && pconv_supported(t, 9, vb.stride, EPI_RELU);
            t.Cin = vb.planes; t.N = 4 * vb.planes; t.resid = &t; packed = packed && pconv_supported(t, 1, 1, EPI_RESID_RELU);
            Ch = 4 * vb.planes;
        }
    }
    auto mtiles = [](long rows) { return (int)((rows + 15) / 16); };
    // backbone_to_vit and the projector on the packed trunk output as well (single-image mode; the ViT pooler keeps the row-major kernels)
    const bool head_packed = packed && !previous && f.v_b2v % 32 == 0 && f.v_proj % 32 == 0 && (4 * f.v_planes[3]) % 32 == 0;
    auto pgemm_rows = [&](const void* X, int K, const GemmW& W, const float* bias, const void* resid, void* out, int rows, int epi, bool rowout) {
        PConvArgs a; memset(&a, 0, sizeof(a));
        a.X = X; a.W = W.w; a.bias = bias; a.resid = resid; a.out = out; a.zero16 = c->zero16;
        a.Hin = a.Win = a.Hout = a.Wout = 1; a.Cin = K; a.N = W.N; a.M = rows; a.mt_in = a.mt_out = mtiles(rows); a.ldo = W.N; a.no_ksplit = c->pconv_noks;
        if (!launch_pconv(dt, a, 1, 1, epi, rowout, s)) c->unsupported = "pconv: a 1 x 1 stride-1 convolution whose input and output row tilings differ";
    };
    if (stem_pool_supported(f.v_stem)) {
        // conv1 + bn1 + relu + maxpool in one launch: the 224^2 x 64 stem output never goes to HBM (stem.hip)
        const long rows0 = (long)B * (S_ / 4) * (S_ / 4);
        if (packed && (rows0 & 15)) {
            // the pad rows of the last 16-row tile are MFMA B columns of every consumer: zero them (once per chunk of 32 channels: that tile is
            // one KiB of [C / 32][tiles][64][8]) instead of feeding whatever the buffer held (448 px has no pad rows, 488 px has)
            const int mt0 = mtiles(rows0);
            for (int kc = 0; kc < f.v_stem / 32; ++kc)
                HIPCHK(c, hipMemsetAsync((char*)c->vbuf[1] + ((size_t)kc * mt0 + (mt0 - 1)) * 1024, 0, 1024, s));
        }
        launch_stem_pool(dt, c->vin, c->v_conv1.w, c->v_conv1_b, c->vbuf[1], B, Hp, Hc, S_ / 4, f.v_stem, packed ? mtiles(rows0) : 0, s);
    } else {
        conv_gemm(c, c->vin, c->v_conv1, c->v_conv1_b, nullptr, c->vbuf[0], B, Hp, Hp, 4, 7, 8, 2, 0, Hc, Hc, EPI_RELU);
        launch_maxpool(dt, c->vbuf[0], c->vbuf[1], B, Hc, Hc, f.v_stem, s);
    }
    Hc = S_ / 4;
    int C = f.v_stem;
    void *cur = c->vbuf[1], *t1 = c->vbuf[0], *t2 = c->vbuf[2], *t3 = c->vbuf[3];
    auto pconv = [&](const void* X, const GemmW& W, const float* bias, const void* resid, void* out, int Hin, int Cin, int taps, int stride, int Hout,
                     int epi, bool rowout) {
        PConvArgs a; memset(&a, 0, sizeof(a));
        a.X = X; a.W = W.w; a.bias = bias; a.resid = resid; a.out = out; a.zero16 = c->zero16;
        a.Hin = Hin; a.Win = Hin; a.Cin = Cin; a.Hout = Hout; a.Wout = Hout; a.N = W.N;
        a.M = B * Hout * Hout; a.mt_in = mtiles((long)B * Hin * Hin); a.mt_out = mtiles(a.M); a.ldo = W.N; a.no_ksplit = c->pconv_noks;
        if (!launch_pconv(dt, a, taps, stride, epi, rowout, s)) c->unsupported = "pconv: a 1 x 1 stride-1 convolution whose input and output row tilings differ";
    };
    for (const VBlock& vb : c->vb) {
        const int Ho = (Hc - 1) / vb.stride + 1;      // 3x3 pad 1 and the 1x1 downsample agree (odd sizes: 61 -> 31)
        if (packed) {
            const bool last = (&vb == &c->vb.back()) && !head_packed;     // row-major NHWC only when the GEMMs behind the trunk are the row-major ones
            pconv(cur, vb.c1, vb.b1, nullptr, t1, Hc, C, 1, 1, Hc, EPI_RELU, false);
            pconv(t1, vb.c2, vb.b2, nullptr, t2, Hc, vb.planes, 9, vb.stride, Ho, EPI_RELU, false);
            const void* idt = cur;
            if (vb.has_ds) { pconv(cur, vb.ds, vb.bds, nullptr, t3, Hc, C, 1, vb.stride, Ho, EPI_NONE, false); idt = t3; }
            pconv(t2, vb.c3, vb.b3, idt, t1, Ho, vb.planes, 1, 1, Ho, EPI_RESID_RELU, last);        // the trunk's output: row-major NHWC
            std::swap(cur, t1);
            Hc = Ho; C = 4 * vb.planes;
            continue;
        }
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
    const int MP = Bimg * P;
    if (head_packed) {
        pgemm_rows(cur, C, c->v_b2v, nullptr, nullptr, t1, MP, EPI_NONE, false);
        pgemm_rows(t1, f.v_b2v, c->v_p1, c->v_p1_b, nullptr, t2, MP, EPI_RELU, false);
    } else
    { GemmArgs a = gargs(cur, C, c->v_b2v, nullptr, t1, f.v_b2v, B * P); run_gemm(c, a, EPI_NONE); }      // [B*P][b2v], NHWC = token order
    if (head_packed) {
    } else if (!previous) {
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
    if (head_packed) pgemm_rows(t2, f.v_proj, c->v_p2, c->v_p2_b, nullptr, t3, MP, EPI_NONE, true);            // row-major [MP][proj] for the scramble / the pooling
    else
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
        return take_unsupported(c);
    }
    // a5: NCHW reshape scramble + ln_vision
    launch_scramble_layernorm(dt, t3, c->v_ln_g, c->v_ln_b, c->v_imgemb, image_embeds, Bimg, P, f.v_proj, f.v_ln_eps, s);
    B = Bimg;

    // a6: Q-Former, query-only path
    const int H = f.q_hidden, NQ = f.q_nquery, M = B * NQ, KVW = c->n_cross * 2 * H;
    { GemmArgs a = gargs(c->v_imgemb, f.v_proj, c->q_wkv, c->q_bkv, c->qkvx, KVW, MP); run_gemm(c, a, EPI_NONE); }
    // Round 4: the hidden state and every GEMM input of the Q-Former fragment-packed (pconv_k as a plain GEMM: both operands straight into
    // registers, the workgroup's waves splitting K; the 1024-row GEMMs of a batch of 32 had 48-192 LDS-staged tiles with 12-48 serial 0.7-us
    // k-steps each, one image's 32 rows ran on the GEMV family): the LayerNorms and the attention output write the packed layout, only
    // Q / K / V stay row-major for the attention kernel.
    bool qpacked = c->trunk_packed && c->zero16 && NQ % 16 == 0 && layernorm_packed_supported(H) && H % 32 == 0 && f.q_inter % 32 == 0;
    if (qpacked) {
        const int mt = M / 16;
        auto pgemm = [&](const void* X, int K, const GemmW& W, const float* bias, const void* resid, void* out, int epi, bool rowout) {
            PConvArgs a; memset(&a, 0, sizeof(a));
            a.X = X; a.W = W.w; a.bias = bias; a.resid = resid; a.out = out; a.zero16 = c->zero16;
            a.Hin = a.Win = a.Hout = a.Wout = 1; a.Cin = K; a.N = W.N; a.M = M; a.mt_in = mt; a.mt_out = mt; a.ldo = W.N; a.no_ksplit = c->pconv_noks;
            if (!launch_pconv(dt, a, 1, 1, epi, rowout, s)) c->unsupported = "pconv: a 1 x 1 stride-1 convolution whose input and output row tilings differ";
        };
        auto attn = [&](const void* Q, long q_ts, const void* K_, const void* V_, long kv_bs, long kv_ts, int Tk) {
            AttnArgs at;
            memset(&at, 0, sizeof(at));
            at.Q = Q; at.q_bs = (long)NQ * q_ts; at.q_ts = q_ts; at.q_hs = 64;
            at.K = K_; at.V = V_; at.k_bs = at.v_bs = kv_bs; at.k_ts = at.v_ts = kv_ts; at.k_hs = at.v_hs = 64;
            at.O = c->qctx; at.o_packed_mt = mt;
            at.B = B; at.H = f.q_heads; at.Tq = NQ; at.Tk = Tk;
            launch_attention(dt, 64, at, s);
        };
        launch_broadcast_packed(dt, c->q_query_ln, c->qx, NQ, H, B, s);
        for (const QLayer& L : c->ql) {
            pgemm(c->qx, H, L.s_wqkv, L.s_bqkv, nullptr, c->qqkv, EPI_NONE, true);
            attn(c->qqkv, 3 * H, (const char*)c->qqkv + (size_t)H * 2, (const char*)c->qqkv + (size_t)2 * H * 2, (long)NQ * 3 * H, 3 * H, NQ);
            pgemm(c->qctx, H, L.s_wo, L.s_bo, c->qx, c->qt, EPI_RESID, false);
            launch_layernorm_packed(dt, c->qt, L.s_g, L.s_b, c->qx, nullptr, M, H, f.q_ln_eps, s);
            if (L.cross_idx >= 0) {
                pgemm(c->qx, H, L.c_wq, L.c_bq, nullptr, c->qqkv, EPI_NONE, true);
                const char* Kx = (const char*)c->qkvx + (size_t)L.cross_idx * 2 * H * 2;
                attn(c->qqkv, H, Kx, Kx + (size_t)H * 2, (long)P * KVW, KVW, P);
                pgemm(c->qctx, H, L.c_wo, L.c_bo, c->qx, c->qt, EPI_RESID, false);
                launch_layernorm_packed(dt, c->qt, L.c_g, L.c_b, c->qx, nullptr, M, H, f.q_ln_eps, s);
            }
            pgemm(c->qx, H, L.w1, L.b1, nullptr, c->qh, EPI_GELU, false);
            pgemm(c->qh, f.q_inter, L.w2, L.b2, c->qx, c->qt, EPI_RESID, false);
            const bool last = (&L == &c->ql.back());
            launch_layernorm_packed(dt, c->qt, L.f_g, L.f_b, last ? nullptr : c->qx, last ? qformer_out : nullptr, M, H, f.q_ln_eps, s);
        }
        HIPCHK(c, hipGetLastError());
        return take_unsupported(c);
    }
    launch_broadcast_rows(dt, c->q_query_ln, c->qx, NQ, H, B, s);
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
    return take_unsupported(c);
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

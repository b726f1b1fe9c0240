// Host-visible launchers of librdx's HIP kernels (internal header; the public C ABI is include/rdx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdx {

enum Epilogue {
    EPI_NONE = 0,        // out = T(acc + bias)
    EPI_RELU = 1,        // out = T(relu(acc + bias))
    EPI_GELU = 2,        // out = T(gelu_erf(acc + bias))
    EPI_RESID = 3,       // out = T(resid + T(acc + bias))
    EPI_SILU_MUL = 4,    // gate/up interleaved tiles: out[:, N/2] = T(T(silu(T(g))) * T(u))
    EPI_LOGITS = 5,      // skinny only: out = T(acc) for n < n_valid, plus per-tile argmax partials
    EPI_RESID_RELU = 6,  // tiled only: out = T(relu(resid + T(acc + bias)))   (Bottleneck tail)
};

struct GemmArgs {
    const void* X; int ldx;          // activations [M][K], row stride in elements
    const void* W;                   // fragment-packed weights (see gemm.hip)
    const float* bias;               // nullable [N]
    const void* resid; int ldr;      // nullable [M][N]
    void* out; int ldo;              // [M][N]  ([M][N/2] for EPI_SILU_MUL)
    int M, N, K;
    const void* norm_w; float eps;   // fused RMSNorm prologue (skinny): weight [K] or null
    float* part_val; int* part_idx;  // EPI_LOGITS: [M][n_tiles]
    int n_valid;                     // EPI_LOGITS: real vocab size (N is padded to 16)
    const int* out_step; long out_step_stride;   // optional: out += (*out_step) * stride elements (per-step score rows)
    // fp8 weights: e4m3 bytes in the 64-deep fragment order (gemm.hip) + one fp32 scale per output row; `W` is null in the engine (no
    // model-dtype copy exists: api_dispatch.hip refuses shapes without an fp8 kernel), only the kernel test hooks set both
    const void* W8; const float* wscale;
    int out_packed;                  // xstat32_k, EPI_SILU_MUL: write the output fragment-packed (input of xsplit32_k)
    int xpacked;                     // xstat32_k: X is the fragment-packed 32-row block written by launch_rmsnorm_packed32 (1 / 2);
                                     // 3 (wstat_k): fragment-packed [k / 32][mtiles][lane][8] over `mtiles` row tiles of 16 (out_packed 3 alike)
    int mtiles;
    int xdup_off;                    // experiments (RDX_XDUP=0): padding rows of a 32-row block load their own (zero / stale) lines instead of re-reading real rows
    // fp8 activations (W8A8: gemm8.hip, xstat32_k<.., A8>): X holds e4m3 bytes, xscale[row][xgroups] their absmax / 448 scales over `xgroups`
    // equal K ranges (1 behind an RMSNorm, 2 o_proj, 4 down_proj); xpacked 4 = the 32-row block in the 64-deep fragment order
    const float* xscale; int xgroups;
    long long* trace;                // debug: [tile][8] timestamps (100 MHz ticks) written by thread 0 of every workgroup (skinny_tile)
};

struct ConvGeom {        // mode 0: plain row-major A.  mode 1: im2col gather from NHWC, K ordered (kh, kw, c)
    int mode;
    int Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad;
};

// fragment-packed convolution / GEMM (pconv.hip): X, resid and out are packed activation tensors [C / 32][mtiles][64 lanes][8]
// (m = NHWC pixel index), W the usual fragment-packed weight with K ordered (kh, kw, c)
struct PConvArgs {
    const void* X; const void* W; const float* bias; const void* resid; void* out;
    const void* zero16;              // >= 16 zero bytes: what taps in the padding read
    int Hin, Win, Cin, Hout, Wout, N;
    int M;                           // output pixels = B x Hout x Wout
    int mt_in, mt_out;               // 16-row tiles of the packed input / output tensors
    int ldo;                         // row stride (elements) when the output is written row-major
    int clog;                        // log2(Cin / 32), filled by launch_pconv
    int no_ksplit;                   // experiments: never let a workgroup's waves split K (rdx_ctx::pconv_noks, RDX_PCONV_KSPLIT=0 at create)
    int tile_m, tile_n, tile_k;      // experiments: forced register tile (MTW x NTW, K split over tile_k waves; 0 = pick; tools/pconv_check.py)
};

struct AttnArgs {        // generic softmax(QK^T/sqrt(D)) V over strided tensors
    const void* Q; const void* K; const void* V; void* O;
    long q_bs, q_ts, q_hs;           // element strides: batch, token, head
    long k_bs, k_ts, k_hs;
    long v_bs, v_ts, v_hs;
    long o_bs, o_ts, o_hs;
    int B, H, Tq, Tk;
    int causal;                      // query i attends keys j <= i + (Tk - Tq)
    int k_perm;                      // K is a decode KV-cache slab in the 16-position fragment order (rdx_common.h kperm), D = 128
    const uint8_t* key_mask; long km_bs;   // nullable [B][>=Tk], 1 = attend
    int flash_min;                   // causal head_dim-128 prefill: take flash_prefill_k from this many 64-query workgroups (0 = never); rdx_ctx::flash_min
    int o_packed_mt;                 // != 0: O is written fragment-packed for wstat_k, [(h D + d) / 32][o_packed_mt][lane][8], row = b Tq + q
};

// Decode-loop state lives in device memory, per batch row (slot_b[b] = KV slot the next token is written to,
// step_b[b] = tokens generated so far, pos[b] = position id of the next token), so that one captured step graph can
// be replayed with identical kernel arguments.

struct LlamaDims {
    int hidden, heads, head_dim, qkv_ld, lora_r;
    float lora_scale;
    int max_len;         // KV slots per (b, head)
    int max_pos;
    int k_perm;          // K cache slabs in the 16-position fragment order (rdx_common.h kperm) instead of row-major
};

void launch_pack_weight(int dtype, const float* src, void* dst, int N, int K, int Npad, const int* rowmap, hipStream_t s);
// per-row absmax e4m3 quantisation: dst8 = fp8 bytes in the 64-deep fragment order, scale[Npad]; dst (nullable; test hooks only) = the
// dequantised weights in the model dtype, standard fragment order (K % 64 == 0)
void launch_pack_weight_fp8(int dtype, const float* src, void* dst8, float* scale, void* dst, int N, int K, int Npad, hipStream_t s);
// skinny GEMM. A fused RMSNorm (a.norm_w != null) is only honoured when skinny_fits_lds(M, K); otherwise the caller
// must normalise first (launch_rmsnorm) and pass norm_w = null.
bool skinny_fits_lds(int M, int K);
void launch_skinny_gemm(int dtype, const GemmArgs& a, int epi, hipStream_t s);
bool xstat32_supported(const GemmArgs& a, int epi);
void launch_xstat32(int dtype, const GemmArgs& a, int epi, hipStream_t s);
// K-split activation-stationary GEMM for the 256-tile projections at 16 < M <= 32: fp32 partial slabs [groups][32][N], combined
// (+ residual, rounding) by the following RMSNorm (launch_rmsnorm_packed32 with `slab`). groups = 0: shape not supported
int xsplit32_groups(const GemmArgs& a);
void launch_xsplit32(int dtype, const GemmArgs& a, float* slab, hipStream_t s);
int xs_min_rows();        // smallest batch on the xstat32 / xsplit32 path (3)
// one prompt's K = 4096 projections in row blocks of 32, the row blocks of a tile walker sharing an XCD's L2 (xstat32_k<.., BLK>): X xpacked 3
bool xstat_blk_supported(const GemmArgs& a, int epi);
void launch_xstat_blk(int dtype, const GemmArgs& a, int epi, hipStream_t s);
// batch 3-16 decode (xs16.hip): activation-stationary K = 4096 projection over ONE row tile with the RMSNorm as its prologue (a.norm_w: X is the
// row-major residual stream; otherwise X is the fragment-packed 32-row block, xpacked 1), and the un-split o_proj / down_proj with the residual
// epilogue (X fragment-packed, xpacked 1; resid / out row-major)
bool xs16_rows_ok(int M);
bool xstat16_supported(const GemmArgs& a, int epi);
void launch_xstat16(int dtype, const GemmArgs& a, int epi, hipStream_t s);
bool xrow16_supported(const GemmArgs& a);
void launch_xrow16(int dtype, const GemmArgs& a, hipStream_t s);
void launch_tiled_gemm(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, hipStream_t s);
// LDS-DMA GEMM for plain row-major activations (M > 32, K % 64 == 0); `ws` = fp32 split-K workspace (nullable)
bool gemm_dma_supported(const GemmArgs& a);
void launch_gemm_dma(int dtype, const GemmArgs& a, int epi, float* ws, size_t ws_floats, hipStream_t s);
// the same kernel with the activation operand gathered from an NHWC tensor (3x3 / strided convolutions, Cin % 8 == 0, K % 64 == 0,
// epilogues NONE / RELU); `zero16` = 16 zero bytes in device memory for taps that fall into the padding
bool gemm_dma_conv_supported(const GemmArgs& a, const ConvGeom& cg, int epi);
void launch_gemm_dma_conv(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, const void* zero16, float* ws, size_t ws_floats,
                          hipStream_t s);

// weight-stationary streaming 1x1 convolution (conv1x1.hip): K <= 256, many rows (the trunk's layer1 / layer2 at batch); cg.mode 0 =
// plain rows, mode 1 with KH = KW = 1 = strided row gather (downsample)
bool conv1x1_stream_supported(const GemmArgs& a, const ConvGeom& cg, int epi);
void launch_conv1x1_stream(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, hipStream_t s);

// many-row GEMM / implicit-GEMM convolution with the weight slice in an LDS ring and the activation fragments fetched straight into
// registers by the wave that owns the rows (wsgemm.hip): K % 64 == 0; convolutions need Cin % 64 == 0
// weight-stationary GEMM for one prompt's prefill: activations fragment-packed (xpacked 3) and streamed past register-resident weights (wstat.hip)
bool wstat_supported(const GemmArgs& a, int epi);
void launch_wstat(int dtype, const GemmArgs& a, int epi, hipStream_t s);
void launch_rmsnorm_packed(int dtype, const void* x, const void* w, void* out, int rows, int mtiles, int H, float eps, hipStream_t s);
void launch_rmsnorm_packed_slab(int dtype, void* x, const void* w, void* out, int rows, int mtiles, float eps, const float* slab, int groups, hipStream_t s);   // H = 4096
// K-split down_proj / o_proj over 33-128 rows (xsplit32_k<.., BLK>): X xpacked 3 over a.mtiles row tiles, slabs [groups][16 mtiles][N]
bool xsplit_blk_supported(const GemmArgs& a);
void launch_xsplit_blk(int dtype, const GemmArgs& a, float* slab, hipStream_t s);
// ... and with fp8 weights (fp8 x fp8): every 32-row block keeps the layouts of the 32-row fp8 kernels at a block stride -- xstat: X = e4m3 blocks (xpacked 4, 32 K bytes
// apart) + xscale[rows]; xsplit: X = model-dtype 64-deep blocks (xpacked 2, 32 K elements apart), slabs [groups][32 blocks][N], groups = 2 (K = 4096) or 4 (K = 11008)
bool xstat_blk8_supported(const GemmArgs& a, int epi);
void launch_xstat_blk8(int dtype, const GemmArgs& a, int epi, hipStream_t s);
int xsplit_blk8_groups(const GemmArgs& a);
void launch_xsplit_blk8(int dtype, const GemmArgs& a, float* slab, hipStream_t s);
void launch_rmsnorm_blk_fp8(int dtype, void* x, const void* w, void* out8, float* xscale, int rows, int mtiles, float eps, const float* slab, int groups, hipStream_t s);
bool wsgemm_supported(const GemmArgs& a, const ConvGeom& cg, int epi);
void launch_wsgemm(int dtype, const GemmArgs& a, const ConvGeom& cg, int epi, const void* zero16, hipStream_t s);

// fragment-packed convolution (pconv.hip): taps 1 | 9 (3 x 3 pad 1), stride 1 | 2, epilogues NONE / RELU / RESID_RELU; rowout = row-major output
bool pconv_supported(const PConvArgs& a, int taps, int stride, int epi);
bool launch_pconv(int dtype, PConvArgs a, int taps, int stride, int epi, bool rowout, hipStream_t s);      // false: refused (1 x 1 stride-1 with mt_in != mt_out), nothing launched
// LayerNorm / broadcast on packed tensors (the Q-Former on packed activations)
bool layernorm_packed_supported(int H);
void launch_layernorm_packed(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* out_f32, int M, int H, float eps, hipStream_t s);
void launch_broadcast_packed(int dtype, const void* src, void* dst, int rows, int H, int B, hipStream_t s);
void launch_pack_rows(int dtype, const void* X, int ldx, void* P, int M, int C, hipStream_t s);      // row-major [M][C] -> packed
void launch_unpack_rows(int dtype, const void* P, void* X, int ldx, int M, int C, hipStream_t s);

// fused ResNet stem (stem.hip): 7x7/2 conv + bias + ReLU + 3x3/2 max pool from the padded NHWC4 image to [B][Ho][Ho][stem]
bool stem_pool_supported(int stem_channels);
// packed_mt != 0: the output is written fragment-packed [stem / 32][packed_mt][64][8] (the layout pconv_k reads) instead of NHWC
void launch_stem_pool(int dtype, const void* in, const void* Wp, const float* bias, void* out, int B, int Hp, int Hc, int Ho, int stem,
                      int packed_mt, hipStream_t s);

void launch_l2_bench(int mode, const void* buf, size_t bytes_per_wg, int shared, int reps, int wgs, unsigned* sink, hipStream_t s);

void launch_attention(int dtype, int head_dim, const AttnArgs& a, hipStream_t s);
// 64-query flash-style blocks for head_dim 128 (flash.hip); launch_attention takes it when the grid fills the chip (RDX_FLASH_MIN workgroups)
bool flash_prefill_supported(int head_dim, const AttnArgs& a);
void launch_flash_prefill(int dtype, const AttnArgs& a, hipStream_t s);
// prefill: LoRA add + RoPE + KV-cache write for T tokens of B rows; q -> qout [B*T][hidden]
void launch_rope_kv_prefill(int dtype, const LlamaDims& d, const void* qkv, const void* lora_bq, const void* lora_bv,
                            const void* cos_t, const void* sin_t, const int* pos_ids, void* qout, void* kcache,
                            void* vcache, int B, int T, int slot0, hipStream_t s);     // rows land at cache slots slot0 + t
void launch_k_unperm(const void* kc, void* out, size_t slabs, int max_len, int perm, hipStream_t s);
// decode: LoRA + RoPE + KV append + attention over the cache for one new token per row
struct DecAttnArgs {
    LlamaDims d;
    const void *qkv, *lbq, *lbv, *cos_t, *sin_t;
    const void* cur_rope = nullptr;  // optional [B][2][128]: cos | sin row of each batch row's CURRENT position (else pos -> tables)
    const int *pos, *slot_b;
    const uint8_t* key_mask;
    void *kcache, *vcache, *out;
    long long* trace = nullptr;      // debug: 8 timestamps (100 MHz ticks) of workgroup (b=0,h=0)
    int out_packed = 0;              // stand-alone launches, batch 3-32: write `out` fragment-packed for xsplit32_k (attn_body.h)
    int out_mt = 2;                  // row tiles of that packed block: 2 (the 32-row block), 3-4 for 33-128 rows ([k / 32][out_mt][lane][8], xpacked 3)
};
void launch_decode_attention(int dtype, const DecAttnArgs& a, int B, hipStream_t s);
// Chained decode launches of the batch <= 2 step (chain.hip): units run as roles of one launch, chained by a fence-free counter
// hand-off (handoff.h) instead of a kernel boundary.
struct ChainLayer { const void *wqkv, *wdown, *attn_norm;
                    // fp8 weights (e4m3, 64-deep fragment order) + per-row scales, null when the model dtype copy is streamed
                    const void *wqkv8, *wdown8; const float *sqkv, *sdown; };
struct ChainArgs {
    const ChainLayer* layers;        // device table, one entry per decoder layer
    int layer;                       // down_proj of this layer, QKV of the next
    int hidden, inter, qkv_n, qkv_ld, B;
    int w8;                          // stream the fp8 weights (every projection quantised)
    float eps;
    void *dx, *dqkv, *dgu;           // [B][hidden] residual stream, [B][qkv_ld], [B][inter]
    int* ctr;                        // chain_ctr_ints(layers) ints, zero at the start of every step
    int* err;
    int naps;                        // poll back-off (x s_sleep(8) between polls)
    int nwg_down;                    // filled by the launcher
};
bool chain_supported(const LlamaDims& d, int inter, int B);
size_t chain_ctr_ints(int layers);
// down_proj(l) (+ residual) -> RMSNorm + QKV(l + 1) in ONE launch, one workgroup per CU; with_next_qkv = false for the last layer
void launch_decode_chain(int dtype, ChainArgs ca, bool with_next_qkv, hipStream_t s);
// decode attention + o_proj (+ residual) in ONE launch of 16-wave workgroups: fast attention body, two o_proj tiles per workgroup with
// their whole K slice in registers, fence-free hand-off. `counter` = 128 ints (8 shards), zero at launch; `err` is set to 1 if a wait
// ever times out (never hangs).
bool attn_oproj16_supported(const LlamaDims& d, int N, int K, int B);
void launch_attn_oproj16(int dtype, const DecAttnArgs& a, const GemmArgs& g, int B, int* counter, int* err, hipStream_t s);

void launch_rmsnorm(int dtype, const void* x, const void* w, void* out, int rows, int H, float eps, hipStream_t s);
// fp8 path (BASELINE configs[4]): e4m3 activations with absmax / 448 scales -- see gemm8.hip for the scheme
void launch_quant_rows(int dtype, const void* x, long ldx, void* out8, float* xscale, int rows, int K, int groups, hipStream_t s);
void launch_rmsnorm_fp8(int dtype, const void* x, const void* w, void* out8, float* xscale, int rows, int H, float eps, hipStream_t s);
void launch_rmsnorm_packed32_fp8(int dtype, void* x, const void* w, void* out8, float* xscale, int rows, int H, float eps, const float* slab,
                                 int groups, hipStream_t s);
// fp8 x fp8 MFMA GEMM over row-major e4m3 activations (any M; N % 16 == 0, K % 64 == 0): epilogues NONE / RESID / SILU_MUL
bool gemm8_supported(const GemmArgs& a, int epi);
void launch_gemm8(int dtype, const GemmArgs& a, int epi, hipStream_t s);
// rows <= 32 normalised into the 32-row fragment-packed block xstat32_k reads (pack 1: 32-deep fragments, 2: fp8 64-deep order)
// slab != null: the rows are first completed as x[row] = x[row] + T(sum over g < groups of slab[g][row][:]) (written back to x)
void launch_rmsnorm_packed32(int dtype, void* x, const void* w, void* out, int rows, int H, float eps, int pack, const float* slab,
                             int groups, hipStream_t s);
void launch_layernorm(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* out_f32,
                      int rows, int H, float eps, hipStream_t s);
void launch_layernorm_ex(int dtype, const void* x, long ldx, const float* gamma, const float* beta, const void* emb, int emb_rows,
                         void* out, long ldo, int rows, int H, float eps, hipStream_t s);
void launch_pool_gather(int dtype, const void* src, void* dst, int B, int L, int C, hipStream_t s);
void launch_pool_concat(int dtype, const void* patch, const void* tokens, void* dst, int B, int L, int C, hipStream_t s);
void launch_scramble_layernorm(int dtype, const void* pp_nhwc, const float* gamma, const float* beta, void* out,
                               float* out_f32, int B, int P, int C, float eps, hipStream_t s);
void launch_img_prep(int dtype, const float* img, void* out, int B, int S, int pad, int Hp, int Wp, hipStream_t s);
void launch_maxpool(int dtype, const void* in, void* out, int B, int H, int W, int C, hipStream_t s);
void launch_broadcast_rows(int dtype, const void* src, void* dst, int rows, int H, int B, hipStream_t s);
void launch_avgpool_flatten(int dtype, const void* in, void* out, int B, int G, int C, int pool, hipStream_t s);
void launch_to_f32(int dtype, const void* src, float* dst, size_t n, hipStream_t s);
void launch_from_f32(int dtype, const float* src, void* dst, size_t n, hipStream_t s);

void launch_prep_prompt(const int* ids, const int* mask_in, int B, int T, int img_id, int pad_id, int* img_pos,
                        int* pos_ids, uint8_t* key_mask, long km_bs, int* pos_next, int* slot_b, int* step_b,
                        int* unfinished, hipStream_t s);
void launch_prep_append(int B, int T_, int keep_len, int* img_pos, int* pos_ids, int* pos_next, int* slot_b, int* step_b,
                        int* unfinished, hipStream_t s);
void launch_embed_splice(int dtype, const int* ids, const int* img_pos, const void* embed, int vocab, const void* img_emb,
                         int n_img, void* out, int B, int T, int H, int use_img, hipStream_t s);
void launch_gather_last(int dtype, const void* x, void* out, int B, int T, int H, hipStream_t s);
void launch_greedy_step(int dtype, const float* part_val, const int* part_idx, int n_tiles, int B, int eos_id, int pad_id,
                        int max_new, int* out_tokens, int* unfinished, int* pos, int* slot_b, int* step_b,
                        const void* embed, int vocab, void* x_next, int H, const int* pos_ro, const void* cos_t,
                        const void* sin_t, void* cur_rope, int* ctr_zero, int n_zero, hipStream_t s);
// pos_ro / cos_t / sin_t / cur_rope: after the update, copy the cos | sin table row of each row's position into
// cur_rope [B][2][128] (pos_ro = the position array even when `pos` is null, i.e. not advanced).
// ctr_zero / n_zero: hand-off counter words to clear for the next decode step (nullable).

// beam search (beam.hip)
#define RDX_MAX_BEAMS 8
void launch_beam_topk(int dtype, const void* logits, const float* beam_scores, int groups, int beams, int vocab, float* cand_score,
                      int* cand_idx, void* logp_out, hipStream_t s);
void launch_kv_beam_reorder(void* kcache, void* vcache, void* scratch, const int* src, const int* start, int rows, int heads, int layers, int max_len,
                            size_t layer_bytes, int p0, int p1, hipStream_t s);
void launch_embed_rows(int dtype, const int* tokens, const void* embed, int vocab, void* x, int rows, int H, hipStream_t s);

}  // namespace rdx

/* librdx -- C ABI of the MI355X-native RaDialog inference path (image encode -> prompt project -> Llama greedy decode).
 *
 * The reference (ChantalMP/RaDialog) has no FFI: its boundary for this path is the Python object API that demo.py /
 * test.py call. librdx sits directly underneath radialog_amd's mirror of that API; each entry point below names the
 * reference interface it replaces. Conventions:
 *   - extern "C", plain pointers and sizes, no torch types. Every data pointer is a DEVICE pointer on the context's
 *     GPU (what torch.Tensor.data_ptr() returns) unless the name ends in _host.
 *   - every function returns 0 on success or a negative error code; rdx_last_error() returns the message. Nothing
 *     throws across the boundary; the library never frees caller memory; outputs are written to caller buffers.
 *   - one rdx_ctx per (process, device); not thread-safe (callers serialise per context). All work is enqueued on
 *     the context's own HIP stream; rdx_sync() blocks until it has drained. Functions that return host-visible
 *     results (rdx_generate's n_steps) synchronise internally.
 */
#ifndef RDX_H
#define RDX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rdx_ctx rdx_ctx;

enum { RDX_DTYPE_F16 = 0, RDX_DTYPE_BF16 = 1 };
enum { RDX_W_GEMM = 0,   /* [rows=N][cols=K] fp32 -> model dtype, re-laid-out into MFMA fragment order            */
       RDX_W_TENSOR = 1, /* [rows][cols] fp32 -> model dtype, row-major (embeddings, norm weights, LoRA B, RoPE)    */
       RDX_W_F32 = 2,    /* [rows][cols] fp32 kept as fp32 (biases, LayerNorm gamma/beta)                           */
       RDX_W_GEMM_FP8 = 3 }; /* quantised to OCP e4m3 with one absmax/448 scale per output row and stored ONLY in that
                            * form (64-deep MFMA fragment order): prefill and batch >= 3 decode multiply fp8 x fp8
                            * (activations e4m3 with a scale per row and K group), batch <= 2 expands the bytes in
                            * registers; a shape without an fp8 kernel makes the call return -8, there is no
                            * dequantised copy to fall back to (BASELINE configs[4]). The 2r LoRA-A rows ride the QKV weight
                            * and are therefore e4m3 too (own row scales, e4m3 input); only LoRA-B is a model-dtype epilogue */

typedef struct rdx_config {
    int dtype;                                     /* RDX_DTYPE_* : arithmetic type of weights/activations         */
    /* Llama decoder (modeling_llama_imgemb.py:433-843; dims of lmsys/vicuna-7b-v1.3) */
    int vocab, hidden, inter, layers, heads, max_pos;
    float rms_eps;
    int lora_r; float lora_scale;                  /* 0 = no adapter; peft LoRA on q_proj,v_proj (finetune.py:167) */
    int qformer_dim;                               /* img_proj_layer input width (demo.py:229)                     */
    /* Q-Former (Qformer.py; blip2.py:47-62) */
    int q_hidden, q_layers, q_heads, q_inter, q_enc_width, q_nquery, q_cross_freq;
    float q_ln_eps;
    /* BioViL-T image encoder (biovil_t/encoder.py:86-136, resnet.py:15-47, model.py:33-91) */
    int v_img, v_stem, v_planes[4], v_blocks[4], v_b2v, v_proj;
    float v_ln_eps;
    /* capacities */
    int max_batch;                                 /* decode rows held in the KV cache: 1 .. 128 (round 5: rows 33-128 decode in 32-row blocks that
                                                    * share an XCD's L2 -- model dtype, or the fp8 x fp8 kernels with fp8 weights, RDX_W_GEMM_FP8)    */
    int max_len;                                   /* KV slots per row (prompt + generated), multiple of 32        */
    int enable_vision, enable_llama;               /* build only one half if 0                                     */
    /* findings classifier (findings_classifier/chexpert_model.py:7-21): the same trunk + projector (v_*) followed by
     * avg_pool2d(cls_pool), fc1 -> ReLU -> fc2. A classifier context sets enable_cls = 1 and enable_vision = 0.    */
    int enable_cls, cls_hidden, cls_classes, cls_pool;
} rdx_config;

/* lifecycle -- replaces model construction (demo.py:149-153 init_blip, :221-236 init_vicuna) */
int rdx_create(rdx_ctx** out, int device_id, const rdx_config* cfg);
void rdx_destroy(rdx_ctx* ctx);
/* 16 hex digits: radialog_amd/build.py source_hash() over the kernel / ABI sources this binary was compiled from ("unstamped" when built
 * without the build script). The Python loader compares it with the hash of the sources next to it and refuses a mismatch. */
const char* rdx_build_hash(void);
const char* rdx_last_error(rdx_ctx* ctx);          /* ctx may be NULL: last error of a failed rdx_create           */
int rdx_sync(rdx_ctx* ctx);
void* rdx_stream(rdx_ctx* ctx);                    /* hipStream_t of the context                                   */

/* weights -- replaces load_checkpoint / from_pretrained / PeftModel.from_pretrained (base_model.py:29-56,
 * demo.py:224-234). `name` is an engine tensor name (radialog_amd/weights.py maps reference state_dict keys to them);
 * `data` is a device fp32 tensor, copied/converted into library-owned storage (the caller may free it afterwards). */
int rdx_set_weight(rdx_ctx* ctx, const char* name, const float* data, int64_t rows, int64_t cols, int kind);
/* The same from a source tensor of another element type: src_dtype RDX_SRC_F32 / RDX_SRC_F16 / RDX_SRC_BF16 (`data` is then a device
 * tensor of that type). A real Vicuna-7B checkpoint is stored in fp16 (from_pretrained(torch_dtype=float16), demo.py:224-226): its
 * tensors go to the library as they are instead of being inflated to fp32 on the host first. fp16 / bf16 sources are widened
 * exactly; the result is bit-identical to rdx_set_weight on the widened tensor. */
enum { RDX_SRC_F32 = 0, RDX_SRC_F16 = 1, RDX_SRC_BF16 = 2 };
int rdx_set_weight_typed(rdx_ctx* ctx, const char* name, const void* data, int src_dtype, int64_t rows, int64_t cols, int kind);
int rdx_finalize_weights(rdx_ctx* ctx);            /* resolves names, checks completeness, allocates workspaces    */

/* image transform -- replaces create_chest_xray_transform_for_inference(resize, center_crop_size)(pil_image) (model/lavis/data/ReportDataset.py:96-106;
 * demo.py:144 / :251: 512 -> 448 for the report model, demo.py:169: 512 -> 488 for the findings classifier) = torchvision Resize(resize) -> CenterCrop(crop)
 * -> ToTensor -> ExpandChannels on the 8-bit "L" image of demo.py:205-218. img: DEVICE uint8 [H][W] row-major; out: DEVICE float32 [3][crop][crop] in [0, 1].
 * Resize on a PIL image is PIL.Image.resize(BILINEAR) (Pillow Resample.c: antialiased triangle filter, two fixed-point passes, uint8 in between): the output
 * equals Pillow's bit for bit. Works on any context (no weights involved). Error: the resized image is smaller than the crop (torchvision would zero-pad). */
int rdx_transform_image(rdx_ctx* ctx, const uint8_t* img, int H, int W, int resize, int crop, float* out);

/* Blip2Qformer.forward_image (blip2_qformer.py:467-484): image float32[B,3,S,S] ->
 *   qformer_out float32[B,n_query,q_hidden] (last_hidden_state), image_embeds float32[B,P,v_proj] (nullable). */
int rdx_encode_image(rdx_ctx* ctx, const float* image, int batch, float* qformer_out, float* image_embeds);
/* Same with a prior study: MultiImageEncoder.forward(current_image, previous_image) (biovil_t/encoder.py:117-123) runs
 * both images through the trunk and takes the difference features from the VisionTransformerPooler
 * (biovil_t/transformer.py:73-224). No RaDialog caller passes a previous image; optional mode of the encoder. */
int rdx_encode_image2(rdx_ctx* ctx, const float* image, const float* previous_image, int batch, float* qformer_out,
                      float* image_embeds);

/* ChexpertClassifier.forward (findings_classifier/chexpert_model.py:15-21; call site demo.py:155-170,:256-261):
 * image float32[B,3,S,S] (S = v_img, 488 in demo.py) -> logits float32[B,cls_classes]; the caller applies sigmoid > 0.5. */
int rdx_classify_findings(rdx_ctx* ctx, const float* image, int batch, float* logits);

/* LlamaForCausalLM.generate(..., num_beams=1) as the reference calls it (demo.py:290-297, test.py:339-348):
 * greedy search (transformers 4.28.1 GenerationMixin.greedy_search) over LlamaForCausalLM.forward with the image
 * splice (modeling_llama_imgemb.py:571-594).
 *   ids          int32[B,T]   left-padded prompt token ids
 *   mask         int32[B,T]   attention mask, or NULL -> ids != pad_id (HF's inferred mask)
 *   qformer_embs float32[B,32,qformer_dim] or NULL (no image splice: `dicom is None and not use_img`)
 *   eos_id       -1 = never stop (throughput runs)
 *   out_tokens   int32[B,max_new]  generated ids; finished rows emit pad_id
 *   scores       model-dtype [max_new][B][vocab] per-step logits (`output_scores=True`), or NULL
 *   n_steps_host host int: number of steps executed (< max_new when every row hit EOS)
 *   use_graph    replay one captured hipGraph per decode step instead of ~160 eager launches */
int rdx_generate(rdx_ctx* ctx, const int32_t* ids, const int32_t* mask, int batch, int T, const float* qformer_embs,
                 int max_new, int eos_id, int pad_id, int32_t* out_tokens, void* scores, int* n_steps_host,
                 int use_graph);

/* The two halves of rdx_generate, for callers that drive the loop themselves (prepare_inputs_for_generation +
 * forward, modeling_llama_imgemb.py:705-836). rdx_prefill runs the prompt and selects token 0; each rdx_decode_step
 * consumes the previously selected token and selects the next. logits: model-dtype [B][vocab] of that step or NULL. */
int rdx_prefill(rdx_ctx* ctx, const int32_t* ids, const int32_t* mask, int batch, int T, const float* qformer_embs,
                int max_new, int eos_id, int pad_id, int32_t* out_tokens, void* logits);
int rdx_decode_step(rdx_ctx* ctx, void* logits);
/* The same step on caller-supplied input ids int32[B] (device) instead of the token the previous step selected: the loop-driving
 * caller's forward(input_ids=[B,1], past_key_values) (modeling_llama_imgemb.py:705-836) -- teacher forcing, constrained decoding. */
int rdx_decode_step_ids(rdx_ctx* ctx, const int32_t* ids, void* logits);

/* Multi-turn re-prompting (test.py:440-674, demo.py:277-305: the reference re-runs the whole conversation each turn). The
 * next turn's prompt usually starts with the previous prompt + answer; its KV rows are still in the cache. These calls keep the
 * first keep_len cache slots of every row (keep_len <= cached positions = previous T + tokens consumed; the caller compares
 * token ids: radialog_amd LlamaForCausalLM.generate does) and run only ids_tail int32[B,T_tail] (no padding, no <IMG>
 * splice) behind them: positions, causal mask and logits are those of the full sequence. Same outputs as rdx_prefill /
 * rdx_generate. */
int rdx_prefill_append(rdx_ctx* ctx, const int32_t* ids_tail, int batch, int T_tail, int keep_len, int max_new, int eos_id,
                       int pad_id, int32_t* out_tokens, void* logits);
int rdx_generate_append(rdx_ctx* ctx, const int32_t* ids_tail, int batch, int T_tail, int keep_len, int max_new, int eos_id,
                        int pad_id, int32_t* out_tokens, void* scores, int* n_steps_host, int use_graph);

/* LlamaForCausalLM.generate(..., num_beams = k > 1) (test.py:467,:629 pass `num_beams=args.num_beams`): transformers 4.28.1
 * GenerationMixin.beam_search driven by BeamSearchScorer (length_penalty 1.0, early_stopping False, one returned hypothesis per
 * prompt by default) over the same forward, with `_reorder_cache` (modeling_llama_imgemb.py:838-843) between steps.
 *   ids / mask / qformer_embs  rows = groups * num_beams, ALREADY expanded the way _expand_inputs_for_generation lays them out
 *                (row r belongs to prompt r / num_beams; the k rows of a prompt are identical); rows <= max_batch
 *   out_tokens_host int32[groups][max_new] (HOST): generated ids of the best hypothesis, pad_id behind its end; the caller appends
 *                the EOS / pads to a common length like BeamSearchScorer.finalize. out_len_host[groups]: generated length;
 *                out_score_host[groups] (nullable): sum of log-probs / length ** length_penalty
 *   step_scores  nullable DEVICE buffer, model dtype [max_new][rows][vocab]: the processed next-token scores (log_softmax) of every
 *                step, what HF returns as `.scores` for beam search
 *   n_steps_host decoder forwards consumed (prompt included) */
int rdx_beam_search(rdx_ctx* ctx, const int32_t* ids, const int32_t* mask, int groups, int num_beams, int T, const float* qformer_embs,
                    int max_new, int eos_id, int pad_id, float length_penalty, int early_stopping, int32_t* out_tokens_host,
                    int32_t* out_len_host, float* out_score_host, void* step_scores, int* n_steps_host);

/* Data-parallel report generation: every image / report is an independent unit, each rank (one process per GPU)
 * runs encode -> prefill -> decode on its own shard with a full weight replica and NO data-path collective; the generated token ids
 * are gathered once at the end with ONE RCCL all-gather over xGMI. The reference has no inference-time collective (its only
 * distributed code is LAVIS' training DDP, runner_base.py:101-118): this is the addition north_star asks for.
 *   rdx_comm_unique_id  rank 0 creates the 128-byte RCCL id (ncclGetUniqueId); the caller hands it to every rank over any host
 *                       channel (radialog_amd/shard.py uses the launcher's TCP store)
 *   rdx_comm_init       ncclCommInitRank on the context's device; collective: every rank of `world` must call it
 *   rdx_allgather_tokens local int32[rows_local][n] (device) -> global int32[world * rows_local][n] (device), rank-major, enqueued
 *                       on the context's stream (rdx_sync to wait); rows_local and n must be the same on every rank */
typedef struct rdx_unique_id { char internal[128]; } rdx_unique_id;
int rdx_comm_unique_id(rdx_unique_id* id_host);
int rdx_comm_init(rdx_ctx* ctx, const rdx_unique_id* id_host, int rank, int world);
int rdx_allgather_tokens(rdx_ctx* ctx, const int32_t* local, int32_t* global, int rows_local, int n);
int rdx_comm_world(rdx_ctx* ctx);                  /* ranks of the initialised communicator, 0 = none */

/* introspection for tests / benchmarks (the kernel-test, trace and microbenchmark hooks are NOT in this library: include/rdx_hooks.h,
 * librdx_hooks.so, loaded under RDX_DEBUG_HOOKS=1 only) */
int rdx_kv_read(rdx_ctx* ctx, int layer, int which /*0=K,1=V*/, void* dst /*model dtype [B][heads][max_len][D]*/);
int rdx_hidden_read(rdx_ctx* ctx, void* dst /*model dtype [B][hidden]: decoder output rows of the last call*/);
/* average duration (ms) of one hot-path unit, measured with HIP events on the context's stream:
 *   what = 0: one decode step (whole graph) at the current state, `iters` replays
 *   what = 1: the gate/up SwiGLU weight-streaming GEMV of every layer in turn, `iters` sweeps -> ms per launch
 *   what = 2: ... the QKV GEMV, 3: o_proj, 4: down_proj, 5: lm_head, 6: decode attention (re-appends the current KV row);
 *   what = 7: the chained down(l) -> QKV(l+1) launch (decode_chain_k, the batch <= 2 default), measured in situ: `iters` eager
 *             decode steps with an event pair around each of its launches (bracket = launch gap + kernel) -> ms per launch
 *   what + 10: the same unit on layer 0 only (weights stay cache resident)
 *   At batch >= 3 the RMSNorm in front of QKV / gate-up / lm_head is a launch of its own: it runs once, outside the timed region, and
 *   units 1, 2, 5 time the GEMM launches alone. */
int rdx_time(rdx_ctx* ctx, int what, int iters, float* ms_host);
/* test / experiment switches of one context (defaults come from the environment at rdx_create): "flash_min" = 64-query workgroups from which the
 * batched prefill attention takes the flash-style kernel (0 never, 1 always; RDX_FLASH_MIN, default 512), "pconv" = the image encoder on
 * fragment-packed activations (1, default) or on the row-major kernels of rounds 1-3 (0; RDX_PCONV), "xs16" = batch 3-16 decode on the one-row-tile
 * family of xs16.hip (1, default) or on the 32-row family (0; RDX_XS16), "prompt_blk" = one prompt's K = 4096 projections on 32-row blocks (1, default) or on
 * the weight-stationary kernel (0; RDX_PBLK). Unknown names return -1. No reference counterpart.
 * At batch 3-16 on the xs16 family rdx_time units 1-5 time THOSE kernels (the RMSNorm is their prologue); max_batch > 32 needs the Vicuna-7B widths
 * (hidden 4096, inter 11008): rdx_finalize_weights refuses other models with more than 32 rows. */
int rdx_set_option(rdx_ctx* ctx, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* RDX_H */

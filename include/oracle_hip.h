/* liboracle_hip -- C ABI of the MI355X-native ORacle multi-view LLaVA inference engine.
 *
 * The reference (egeozsoy/ORacle) has no FFI for this path: its boundary is a Python API
 * (SURVEY.md §8b). Every entry point below therefore names the reference Python function whose GPU work
 * it replaces; the host-side mirror in oracle_amd/llava/ keeps those functions' signatures and calls
 * these entry points through ctypes (binding shown in INTEGRATION.md).
 *
 * Conventions: extern "C"; plain pointers and sizes; return 0 on success, negative on error with a
 * message in orc_last_error(); caller owns every buffer it passes (device pointers unless marked host);
 * the engine owns its weight copies, KV cache and workspace; every call is asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = default stream) unless noted; one engine is not thread-safe.
 * All activations / weights are bf16 (uint16 storage), logits fp32, ids int32.
 */
#ifndef ORACLE_HIP_H
#define ORACLE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_engine orc_engine;

typedef struct orc_config {
  /* Vicuna/LLaMA (LlavaConfig; reference LLaVA/llava/model/language_model/llava_llama.py:27-46) */
  int32_t vocab, d_model, n_layers, n_heads, d_ff;
  float rms_eps, rope_theta;
  /* CLIP vision tower (reference multimodal_encoder/clip_encoder.py:22-27) */
  int32_t clip_layers, clip_d, clip_heads, clip_ff, clip_image, clip_patch;
  float clip_eps;
  /* learned multi-view pooler (reference multimodal_projector/builder.py:56-72) */
  int32_t pool_layers, pool_heads, pool_ff, pool_maxpos;
  float pool_eps;
  /* capacities */
  int32_t max_batch;      /* KV-cache slots (<= 64); one prefill / decode launch runs over at most 32 of them (rows) */
  int32_t max_seq;        /* KV-cache slots per row: spliced prompt + new tokens */
  int32_t max_images;     /* images per vision call */
  int32_t max_new_tokens; /* capacity of the on-device generated-token buffer */
  int32_t pad_token_id, eos_token_id;
} orc_config;

const char* orc_last_error(void);
int orc_abi_version(void);

/* load_pretrained_model (reference LLaVA/llava/model/builder.py:26-185): weight ingest. */
int orc_engine_create(const orc_config* cfg, orc_engine** out);
void orc_engine_destroy(orc_engine* e);
/* `name` is the reference checkpoint key (e.g. "model.layers.0.self_attn.q_proj.weight"); data is bf16,
 * row-major, `shape[ndim]`; on_device != 0 means `data` is a device pointer. Synchronous. */
int orc_engine_load_weight(orc_engine* e, const char* name, const void* data, int on_device,
                           const int64_t* shape, int ndim);
/* packs weights into MFMA fragment order, builds rotary tables, allocates KV cache + workspace. Synchronous. */
int orc_engine_finalize(orc_engine* e);

/* CLIPVisionTower.forward + feature_select (reference clip_encoder.py:29-51):
 * images (n,3,I,I) -> hidden_states[-2][:,1:] (n, patches, clip_d). */
int orc_vision_encode(orc_engine* e, const void* images, int n_images, void* out_features, void* stream);
/* CLIPVisionTower.global_forward (reference clip_encoder.py:53-64): pooler_output (n, clip_d). */
int orc_vision_global(orc_engine* e, const void* images, int n_images, void* out_pooled, void* stream);
/* pad_embeddings + ImageEmbeddingPooler.forward (reference llava_arch.py:143-183, projector/builder.py:74-82):
 * features (sum(views), patches, clip_d) grouped per row by views_per_row (host, B ints)
 * -> (B, patches, clip_d). */
int orc_pool_views(orc_engine* e, const void* features, const int32_t* views_per_row_host, int batch,
                   void* out_pooled, void* stream);
/* mm_projector mlp2x_gelu (reference multimodal_projector/builder.py:41-48): (rows, clip_d) -> (rows, d_model). */
int orc_project(orc_engine* e, const void* x, int rows, void* out, void* stream);
/* multi-view reductions of llava_arch.py:216-227 (mode 0 max, 1 mean): (B,V,n) -> (B,n). */
int orc_view_reduce(orc_engine* e, const void* x, int batch, int views, int64_t n, int mode, void* out, void* stream);
/* encode_images_concat_pool (reference llava_arch.py:128-141): 2x2 max pool of the token grid. */
int orc_pool2x2(orc_engine* e, const void* x, int n_images, int grid, int d, void* out, void* stream);

/* process_images with image_aspect_ratio == 'pad' (reference LLaVA/llava/mm_utils.py:14-40) + the square-image part of
 * CLIPImageProcessor.preprocess (third-party transformers; PIL bicubic resize, /255, mean/std) + the bf16 cast of
 * scene_graph_prediction_model_oracle.py:65-67: n decoded RGB views of one size, uint8 HWC on the device ->
 * (n, 3, out_size, out_size) bf16. bounds (out_size x 2 int32: first tap, tap count) and coeffs (out_size x ksize int32,
 * 22-bit fixed point) are Pillow's resample tables for max(height, width) -> out_size; lut_bf16 is the (3, 256)
 * normalisation table; fill_rgb_host the expand2square colour (3 ints, host); tmp >= n * max(height, width) *
 * out_size * 3 bytes of device scratch. Integer arithmetic: bit-exact with PIL + numpy float32. No engine needed. */
int orc_preprocess_images(const uint8_t* rgb_hwc, int n_images, int height, int width, int out_size,
                          const int32_t* bounds, const int32_t* coeffs, int ksize, const void* lut_bf16,
                          const int32_t* fill_rgb_host, void* tmp, void* out_pixel_values, void* stream);

/* No counterpart in the reference: orc_preprocess_images for callers that cannot allocate or bring a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc). Same arguments without `tmp`; the image between
 * the two resample passes lives in a static per-GPU device buffer of orc_preprocess_noalloc_capacity() bytes (16 MiB, held
 * until the process ends), so n_images * max(height, width) * out_size * 3 must not exceed it (error otherwise). Calls no
 * hipMalloc / hipFree. The first call on a device resolves that device's buffer address (a call with n_images == 0 does
 * only that: make it once outside graph capture). All calls on one GPU share the buffer: calls that overlap on one GPU
 * (different streams or threads) race. */
int orc_preprocess_images_noalloc(const uint8_t* rgb_hwc, int n_images, int height, int width, int out_size,
                                  const int32_t* bounds, const int32_t* coeffs, int ksize, const void* lut_bf16,
                                  const int32_t* fill_rgb_host, void* out_pixel_values, void* stream);
int64_t orc_preprocess_noalloc_capacity(void);

/* prepare_inputs_labels_for_multimodal, step 0 (reference llava_arch.py:245-359): gather the spliced
 * (rows, d_model) embedding matrix. codes[r] = kind<<28 | index; kind 0 = zero (padding) row,
 * 1 = embed_tokens[index], 2 = image_features[index], 3 = descriptors[index]. */
int orc_splice(orc_engine* e, const int32_t* codes, int rows, const void* image_features, const void* descriptors,
               void* out_embeds, void* stream);

/* LlavaLlamaForCausalLM.forward, step 0 of generate() (reference llava_llama.py:54-100): prefill over the
 * left-padded (batch, T, d_model) embeddings; fills the KV cache; resets the greedy-search state; samples
 * the first token. position_ids (batch*T, device); left_pad_host[b] = number of padding rows of row b.
 * logits_out: optional device (batch, vocab) fp32 copy of the last-position logits. */
int orc_prefill(orc_engine* e, const void* embeds, const int32_t* position_ids, const int32_t* left_pad_host,
                int batch, int T, float* logits_out, void* stream);
/* one KV-cached greedy step (reference llava_arch.py:192-201 + llava_llama.py:54-112 + greedy search):
 * feeds forced_ids (device, batch ints) or, when NULL, the tokens sampled by the previous step. */
int orc_decode_step(orc_engine* e, const int32_t* forced_ids, float* logits_out, void* stream);
/* runs up to n_steps decode steps back to back on the stream (no host round trip). */
int orc_decode_steps(orc_engine* e, int n_steps, void* stream);
/* synchronises `stream` and reads the greedy-search state: tokens_host (batch * max_new_tokens, row-major),
 * number of steps taken so far, rows still unfinished, raw argmax of the last step (batch ints, optional). */
int orc_read_state(orc_engine* e, int32_t* tokens_host, int32_t* steps_host, int32_t* n_unfinished_host,
                   int32_t* last_argmax_host, void* stream);

/* per-row length stops of the running batch (after orc_prefill; default = max_new_tokens for every row): row b is finished
 * once it has emitted limits_host[b] tokens (1..max_new_tokens; the first token was sampled by the prefill itself, so a row with
 * limit 1 is marked finished by this call) or EOS, whichever comes first -- what max_new_tokens is to the whole batch in the reference's generate() call
 * (scene_graph_prediction_model_oracle.py:135-143), per row; used for synthetic answer-length distributions. */
int orc_set_row_limits(orc_engine* e, const int32_t* limits_host, int n, void* stream);

/* Continuous batching over the reference's outer loop (`for batch in dataloader: self.forward(batch)`,
 * scene_graph_prediction_model_oracle.py:251-260): frames are independent, so a decode row whose frame has finished (EOS or
 * its length stop) takes over the next frame while the other rows keep decoding. The engine has max_batch KV-cache SLOTS,
 * each with its own greedy state on the device (cached length, left pad, step counter, length stop, unfinished flag,
 * next token, token buffer); a decode launch runs over n_rows ROWS (n_rows <= min(max_batch, 32)), each attached to one slot.
 *   orc_stream_begin   n_rows decode rows, none attached.
 *   orc_prefill_slots  LlavaLlamaForCausalLM.forward step 0 (llava_llama.py:54-100) for n left-padded rows (embeds
 *                      (n, T, d_model), position_ids (n*T) device, left_pad_host[n]) into the UNATTACHED slots
 *                      slots_host[n]: fills their KV cache, resets their state with the length stops limits_host[n]
 *                      (<= max_new_tokens, T + limit <= max_seq), samples each row's first token. Batched like orc_prefill,
 *                      so a whole incoming batch can be prefilled AHEAD into free slots while other frames decode.
 *   orc_stream_attach  host-only: decode row `row` continues with slot `slot` (-1 detaches) from the next decode step on.
 *   orc_decode_steps   advances every attached, unfinished row; a finished row freezes (no cache growth, no cache reads).
 *   orc_read_slots     synchronises and reads, for all max_batch slots: tokens (max_batch * max_new_tokens, pad-filled),
 *                      tokens emitted, unfinished flags.
 * orc_prefill (a whole batch into slots 0..B-1, rows attached 1:1) ends the stream. */
int orc_stream_begin(orc_engine* e, int n_rows, void* stream);
int orc_prefill_slots(orc_engine* e, const int32_t* slots_host, int n, const void* embeds, const int32_t* position_ids,
                      const int32_t* left_pad_host, const int32_t* limits_host, int T, float* logits_out, void* stream);
int orc_stream_attach(orc_engine* e, int row, int slot);
int orc_read_slots(orc_engine* e, int32_t* tokens_host, int32_t* row_steps_host, int32_t* unfinished_host, void* stream);

/* test / profiling hooks for single kernels (same kernels the engine runs). */
int orc_test_gemm(const void* x, int M, int K, const void* w_rowmajor, int N, const void* bias, const void* residual,
                  int epilogue, int use_gemv, void* out, void* stream);
int orc_test_rmsnorm(const void* x, const void* w, int rows, int d, float eps, void* out, void* stream);
/* RMSNorm -> Linear of the decode path, both ways (LlamaDecoderLayer's input_layernorm -> q/k/v_proj etc. under
 * LLaVA/llava/model/language_model/llava_llama.py:87-98): out_exact = Linear(rmsnorm(x)) with the reference's rounding points,
 * out_folded = the engine's default decode arithmetic (norm weight folded into the weight copy, row scale in the GEMV epilogue;
 * ORC_DECODE_NORM=exact switches the engine to the first form). x (M <= 32, K) bf16, w_norm (K), w (N, K); outputs (M, N) bf16. */
int orc_test_folded_norm(const void* x, int M, int K, const void* w_norm, const void* w, int N, float eps, void* out_exact,
                         void* out_folded, void* stream);
int orc_test_layernorm(const void* x, const void* w, const void* b, int rows, int d, float eps, void* out, void* stream);
int orc_test_attention(const void* q, const void* k, const void* v, int B, int H, int Sq, int Sk, int head_dim,
                       const int32_t* kbeg, const int32_t* kend, int causal, int q_off, float scale, void* out,
                       void* stream);
/* decode attention over a KV cache, the engine's own launch (un-split for batch * heads >= 192, split + combine below):
 * q (batch, heads*128); kcache / vcache (batch, heads, t_max, 128); kbeg device (batch) first valid key per row or NULL;
 * keys [kbeg[b], n_keys) are attended (the decode-step branch of LlamaAttention under llava_llama.py:54-100). Synchronous. */
int orc_test_decode_attention(const void* q, const void* kcache, const void* vcache, int batch, int heads, int t_max,
                              const int32_t* kbeg, int n_keys, void* out, void* stream);
/* HIP-event sampling of the dominant decode kernel (gate/up projection GEMV of every 8th layer): start / stop events
 * attached to the launch itself (the dispatch's timestamps) on the stream the kernel is launched on. enable(1) resets the samples; read() synchronises the device and returns
 * the summed duration, the number of sampled launches and the algorithmic bytes of one launch. */
int orc_profile_enable(orc_engine* e, int on);
int orc_profile_read(orc_engine* e, double* total_ms, int32_t* count, int64_t* bytes_per_launch);
/* ALGORITHMIC bytes of weights of one decode step (bf16: 2 N K per projection) / bytes of KV read per cached key per row
 * (roofline bookkeeping). */
int64_t orc_decode_weight_bytes(orc_engine* e);
int64_t orc_kv_bytes_per_key(orc_engine* e);
/* PHYSICAL weight bytes the decode kernels stream: the engine keeps lossless 12-bit copies of the decode weights (768 B per
 * 1-KiB fragment block + 1 B per block, bit-exact decode in the GEMV; environment ORC_DECODE_WEIGHTS=plain turns it off).
 * which = 0: one whole decode step; which = 1: one gate/up launch (the kernel orc_profile_read samples). No counterpart in the
 * reference (measurement only). */
int64_t orc_decode_stream_bytes(orc_engine* e, int which);
/* Bookkeeping of those 12-bit copies, out6 = {decode weight matrices, matrices that took the 12-bit stream, weight elements,
 * escaped elements (patch-table entries), 1-KiB blocks, blocks with at least one escape}. A matrix with a GEMV unit of more than
 * 254 non-zero out-of-range elements keeps bf16: orc_engine_finalize says so on stderr, once per matrix and once in total.
 * No counterpart in the reference (measurement only). */
int orc_decode_weight_stats(orc_engine* e, int64_t* out6);

#ifdef __cplusplus
}
#endif
#endif

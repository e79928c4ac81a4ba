// Internal launcher interface between the engine (engine.hip) and the kernel files. Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

// records "what: detail" for orc_last_error() and returns -1 (engine.hip)
int orc_fail(const char* what, const char* detail);

// Row -> slot map, passed by value inside the kernel arguments: row b of a launch (a decode row, or a row of a prefill
// batch) uses the per-slot greedy state and the KV-cache rows of slot[b]; -1 = inactive row (nothing read or written).
// The host changes the map between launches without touching device memory (continuous batching: a finished frame's
// decode row is pointed at a slot that was prefilled ahead of time).
constexpr int kMaxRows = 32;    // rows of one decode / prefill launch (two MFMA B-operand tiles of 16 in the decode GEMVs)
struct RowMap {                 // default: row b <-> slot b
  int slot[kMaxRows] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31};
};
inline RowMap identity_map() { return RowMap(); }

enum { EPI_NONE = 0, EPI_QUICKGELU = 1, EPI_GELU = 2, EPI_RESIDUAL = 3, EPI_SWIGLU = 4, EPI_F32 = 5, EPI_ROPE_KV = 6 };

struct GemmArgs {
  const bf16_t* X; int ldx;   // activations [M][K] row-major
  const bf16_t* Wp;           // packed weights, Np = N rows (multiple of 128), Kp = K (multiple of 64)
  const bf16_t* bias;         // [N] or null
  const bf16_t* R; int ldr;   // residual [M][N] (EPI_RESIDUAL)
  void* C; int ldc;           // bf16 (fp32 for EPI_F32); EPI_SWIGLU writes N/2 columns
  int M, N, K;
  int Nvalid;                 // columns actually stored (<= N, or <= N/2 for SWIGLU)
};
hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
void gemm_force_kernel(int which);   // tools only: 0 auto, 1 = 128^2 register-staged, 2 / 3 = LDS-DMA kernel BM 256 / 128
// decode-time skinny GEMM (M <= 16), same argument meaning; N multiple of 16 (32 for SWIGLU)
hipError_t launch_gemv(const GemmArgs& a, int epi, hipStream_t s);
// full argument set of the decode GEMV (fused rotary + KV-cache epilogue, sum-of-squares partials)
struct GemvArgs {
  const bf16_t* X; int ldx;
  const bf16_t* Wp;
  const bf16_t* bias;
  const bf16_t* R; int ldr;
  void* C; int ldc;
  int M, N, K, Nvalid;
  // EPI_RESIDUAL: when non-null, ssq_out[b * (N/16) + ntile] = sum of squares of the stored tile values of row b
  float* ssq_out;
  // EPI_ROPE_KV: W rows packed in units of 3 tiles: (tile j, tile j+4) of a q or k head, then one v tile; C = q output [M][H*128]
  int H; const int* kbeg; const int* cur_len;   // per SLOT: first valid key; cached keys incl. the token of this step
  RowMap map;                                   // decode row -> slot (state and cache rows)
  const bf16_t* cos_tab; const bf16_t* sin_tab; int maxpos;
  bf16_t* Kc; bf16_t* Vc; long c_bs, c_hs;
  // EPI_ROPE_KV, optional (null: the epilogue chases map -> cur_len / kbeg -> cos / sin itself): what launch_embed_tokens left per
  // decode ROW for this step -- row_meta[4 b] = {slot or -1, cached keys incl. this token, left pad, unfinished}, row_cos / row_sin
  // [b][64] = the rotary table rows of the token's position. Read at kernel entry by the pipelined kernel.
  const int* row_meta = nullptr; const bf16_t* row_cos = nullptr; const bf16_t* row_sin = nullptr;
  // decode-path activation buffers in the fragment-ordered layout (common.h xf_off): x_xf = X is, c_xf = C is written so
  // (EPI_SWIGLU only: its output feeds the down projection)
  int x_xf, c_xf;
  // RMSNorm folded into the projection (decode path): X holds the RAW hidden rows, the weight copy carries the norm weight
  // (W'[n][k] = bf16(W[n][k] * w_norm[k]), packed at finalize) and, when ssq_in is non-null, every epilogue multiplies
  // row b by r_b = rsqrt(sum_t ssq_in[b][t] / norm_d + norm_eps) before its first rounding: y = bf16(r_b * sum_k W' h).
  // ssq_in = the per-n-tile partials the producer of the hidden state left (ssq_out of the residual epilogue / the token gather).
  const float* ssq_in; int n_part; int norm_d; float norm_eps;
  // EPI_RESIDUAL: when non-null the new hidden rows are ALSO written in the fragment-ordered layout (input of the next projection)
  bf16_t* C_xf;
  // Losslessly compressed weights (gemv.hip "12-bit blocks"): when Wc is non-null the kernel streams Wc / emin instead of Wp.
  // Wc: per 1-KiB fragment block 768 B (lane l: 12 B = its 8 elements at 12 bits: sign, 4-bit exponent offset, 7-bit mantissa);
  // emin[block]: bits 0-6 = (smallest coded exponent field) / 2, bit 7 = the block has escaped elements;
  // patch: true bf16 bits of the escaped elements, 256 entries per GEMV unit
  const uint8_t* Wc = nullptr; const uint8_t* emin = nullptr; const uint16_t* patch = nullptr;
  // order of the 768-B blocks inside Wc (cw_block_index, gemv.hip): 0 = row-major, block (ntile, kt) at ntile * KT + kt; G > 0 = interleaved in
  // groups of G k-tiles -- all n-tiles' blocks of k-tiles [g G, (g + 1) G) lie together -- so that the workgroups of a launch, which walk
  // K in step, read ONE contiguous window of the matrix at a time instead of N / 16 regions a fixed stride apart. emin / patch keep the
  // row-major block numbering.
  int wc_group = 0;
  // optional (host side only): HIP events that bracket exactly this kernel's execution (hipExtLaunchKernelGGL start /
  // stop events = the dispatch's own timestamps, what rocprofv3's kernel trace reports), for orc_profile_read
  void* ev_start; void* ev_stop;
  // tools/decode_bench only (gemv_pipe_kernel): timing experiments that keep the instruction stream and drop the memory requests --
  // bit 0: the activation descriptor has zero records, bit 1: the weight descriptor has (results are garbage)
  int dbg = 0;
  // bit 2 (GEMV_ALL_VARIANTS builds): wave 0 of every workgroup writes s_memrealtime stamps (100 MHz) to dbg_ts[block][4]:
  // kernel entry, first weight fragment landed, k-loop done, epilogue done
  unsigned long long* dbg_ts = nullptr;
};
hipError_t launch_gemv_ex(const GemvArgs& a, int epi, int variant, hipStream_t s);
void gemv_force_legacy(int on);   // tests / tools: 1 = run the r2 kernels (branchy k-loop) instead of the pipelined ones; same bits
// lossless 12-bit compression of a packed matrix for the decode GEMVs (gemv.hip): Wc N*K*3/2 bytes, emin N/16*K/32 bytes (+3 pad),
// patch (N/16/nt_unit)*256 uint16, cnt N/16/nt_unit ints; *overflow_dev is set when a unit has more than 254 non-zero escapes (escaped zeros share two constant entries)
int gemv_unit_tiles(int epi, int N);
constexpr int kCwGroup = 16;   // the decode GEMVs' window: 8 waves x 2 k-tiles or 16 waves x 1 per workgroup step
hipError_t launch_compress_weight(const bf16_t* Wp, int N, int K, int nt_unit, uint8_t* Wc, uint8_t* emin, uint16_t* patch, int* cnt,
                                  int* overflow_dev, hipStream_t s, int group = kCwGroup);
// colscale (bf16 [K], may be null): Wp holds bf16(W[n][k] * colscale[k])
hipError_t launch_pack_weight(const bf16_t* W, int Nsrc, int K, const int* rowmap, bf16_t* Wp, int Np, int Kp,
                              hipStream_t s, const bf16_t* colscale = nullptr);

// ---- norms
hipError_t launch_rmsnorm(const bf16_t* x, int ldx, const bf16_t* w, bf16_t* y, int ldy, int rows, int d, float eps,
                          hipStream_t s);
// RMSNorm with the row sums of squares supplied as n_part partials per row (part[row][n_part])
// y_xf: y is written in the fragment-ordered layout (common.h xf_off) instead of [row][ldy]
hipError_t launch_rmsnorm_part(const bf16_t* x, int ldx, const float* part, int n_part, const bf16_t* w, bf16_t* y, int ldy,
                               int rows, int d, float eps, hipStream_t s, int y_xf = 0);
hipError_t launch_layernorm(const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int ldy, int rows,
                            int d, float eps, hipStream_t s);

// ---- attention (flash style, MFMA). Generic strides in elements.
struct AttnArgs {
  const bf16_t* Q; long q_bs, q_hs, q_ts;   // batch / head / token strides
  const bf16_t* K; long k_bs, k_hs, k_ts;
  const bf16_t* V; long v_bs, v_hs, v_ts;
  bf16_t* O; long o_bs, o_hs, o_ts;
  int B, H, Sq, Sk;
  const int* kbeg;   // [B] first valid key (null = 0)
  const int* kend;   // [B] one past last valid key (null = Sk)
  int causal;        // key j visible to query i iff j <= i + q_off
  int q_off;
  float scale;
  int kv_map_on = 0; // 1: K / V batch index of row b is kv_map.slot[b] (prefill into arbitrary KV-cache slots)
  RowMap kv_map;
};
hipError_t launch_attention(const AttnArgs& a, int head_dim, hipStream_t s);

// ---- decode attention over the KV cache (one query token per row)
struct DecodeAttnArgs {
  const bf16_t* Q; long q_bs;                // [B][H*128]
  const bf16_t* Kc; const bf16_t* Vc;        // [B][H][Tmax][128]
  long c_bs, c_hs;
  bf16_t* O; long o_bs;                      // [B][H*128]
  float* part;                               // workspace [B][H][NSPLIT][2+128]
  const int* kbeg;                           // [B] left-pad offset
  const int* cur_len;                        // per slot: number of cached keys incl. the new one
  const int* unfinished = nullptr;           // per slot, or null: rows whose flag is 0 are skipped (output zeros; their tokens are pad)
  RowMap map;                                // row -> slot (kbeg / cur_len / unfinished / cache rows); Q, O and part are per row
  int B, H;
  float scale;
  int o_xf;                                  // write O in the fragment-ordered layout (common.h xf_off, 16-row images), k = h * 128 + d
  // tools/decode_bench (GEMV_ALL_VARIANTS builds): s_memrealtime stamps per workgroup: [block][4] = entry, lengths known, key loop
  // done (wave 0), end; then [block][16] per-wave (entry -> key loop done)
  unsigned long long* dbg_ts = nullptr;
};
#define DECODE_NSPLIT 8   // capacity of the split-KV partial buffer; the launch picks 2..8 splits by batch
hipError_t launch_decode_attention(const DecodeAttnArgs& a, hipStream_t s, int variant = 0);

// ---- rotary + KV-cache write. qkv [rows][3*H*128]; row = b*T + t; q rotated in place.
struct RopeArgs {
  bf16_t* qkv; int B, T, H;
  const int* pos;             // [B*T] position ids (null: decode mode, pos = cur_len[b] - 1 - kbeg[b])
  const int* kbeg;            // decode mode only
  const int* cur_len;         // decode mode: [B]; cache slot = cur_len[b] - 1
  const bf16_t* cos_tab; const bf16_t* sin_tab;   // [maxpos][64]
  int maxpos;
  bf16_t* Kc; bf16_t* Vc; long c_bs, c_hs;        // cache [slot][H][Tmax][128]
  int t0;                     // prefill: cache position of token t is t0 + t
  RowMap map;                 // batch row -> cache slot
};
hipError_t launch_rope_kv(const RopeArgs& a, hipStream_t s);

// ---- small data-movement kernels
hipError_t launch_im2col(const bf16_t* img, int n_img, int image, int patch, bf16_t* out, int kp, hipStream_t s);
hipError_t launch_clip_embed(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, const bf16_t* lnw,
                             const bf16_t* lnb, bf16_t* out, int n_img, int tokens, int d, float eps, hipStream_t s);
hipError_t launch_bert_embed(const bf16_t* feats, const int* row_off, const int* row_valid, const bf16_t* tt0,
                             const bf16_t* pos, const bf16_t* lnw, const bf16_t* lnb, bf16_t* out, int B, int S, int d,
                             float eps, hipStream_t s);
hipError_t launch_strip_cls(const bf16_t* h, bf16_t* out, int n_img, int tokens, int d, hipStream_t s);
hipError_t launch_copy_rows(const bf16_t* src, long src_bs, bf16_t* dst, long dst_bs, int B, int rows, int d,
                            hipStream_t s);
// splice: out[r][:] = kind==0 ? 0 : table_kind[idx][:]   (code = kind<<28 | idx)
hipError_t launch_gather_rows(const int* codes, const bf16_t* tok, const bf16_t* img, const bf16_t* desc, bf16_t* out,
                              int rows, int d, hipStream_t s);
// out[r] = tok[id_r], id_r = ids_by_row[r] when given (teacher forcing), else ids_by_slot[map.slot[r]] (inactive row: token 0);
// when ssq_part != null also part[r][0] = sum(out[r]^2), part[r][1..n_part) = 0; when out_xf != null the rows are also
// written in the fragment-ordered layout (rows <= 32: two 16-row images)
// meta (optional): per-row operands of this decode step's q/k/v epilogues (GemvArgs::row_meta / row_cos / row_sin), gathered here once
struct RowMetaArgs {
  const int* cur_len = nullptr; const int* kbeg = nullptr; const int* unfinished = nullptr;   // per slot
  const bf16_t* cos_tab = nullptr; const bf16_t* sin_tab = nullptr; int maxpos = 0;
  int* row_meta = nullptr; bf16_t* row_cos = nullptr; bf16_t* row_sin = nullptr;              // outputs: [rows][4], [rows][64], [rows][64]
};
hipError_t launch_embed_tokens(const int* ids_by_row, const int* ids_by_slot, const RowMap& map, const bf16_t* tok, bf16_t* out,
                               int rows, int d, int vocab, float* ssq_part, int n_part, hipStream_t s, bf16_t* out_xf = nullptr,
                               const RowMetaArgs* meta = nullptr);
hipError_t launch_view_reduce(const bf16_t* x, bf16_t* out, int B, int V, int n, int mode, hipStream_t s);
hipError_t launch_pool2x2(const bf16_t* x, bf16_t* out, int n_img, int grid, int d, hipStream_t s);

// ---- image preprocessing: pad to square + Pillow-exact 8-bit bicubic resize + normalisation table (misc.hip)
hipError_t launch_preprocess(const uint8_t* img, int n, int H, int W, int out, const int* bounds, const int* coef, int ksize,
                             const bf16_t* lut, int fr, int fg, int fb, uint8_t* tmp, bf16_t* dst, hipStream_t s);

// ---- greedy sampling state machine (device resident, per row: rows of one batch may belong to different frames that
// started at different times -- continuous batching, orc_prefill_slot)
struct SampleArgs {
  const float* logits; int vocab;
  int B;               // rows sampled by this launch: logits row i belongs to slot map.slot[i] (-1: skipped)
  RowMap map;
  RowMap count; int n_count;   // slots whose unfinished flags make up n_unfinished (the rows being decoded)
  // per-slot state:
  int* unfinished;
  int* next_ids;       // token fed to the next step
  int* raw_argmax;     // argmax before the finished-row substitution (may be null)
  int* out_tokens; int out_ld;   // [slot][max_new] generated tokens (pre-filled with pad at prefill)
  int* row_step;       // tokens emitted so far by the frame in this slot
  int* cur_len;        // cached keys for the NEXT step incl. its token
  const int* limit;    // length stop: the frame is finished after emitting limit[slot] tokens
  int* step;           // device scalar: sampling launches since the last batch prefill
  int* n_unfinished;   // device scalar: rows still unfinished after this step
  int pad_id, eos_id;
  int freeze_finished; // 1: a finished row stops advancing (keeps its cache slot and step); 0: teacher forcing, all rows advance
};
hipError_t launch_argmax_sample(const SampleArgs& a, hipStream_t s);

// ---- per-slot greedy state (re)initialisation for a prefill: one block per prefilled row, everything by value
struct SlotInitArgs {
  RowMap map; int k;                 // prefill row i -> slot
  int kbeg[kMaxRows]; int limit[kMaxRows];   // per prefill row: left pad, length stop
  int T; int pad_id; int out_ld;
  int *unfinished, *kbeg_state, *cur_len, *row_step, *limit_state, *tokens;   // per-slot state arrays
  int* pf_kbeg;                      // [kMaxRows] per prefill ROW copy of kbeg (read by the prefill attention)
};
hipError_t launch_slot_init(const SlotInitArgs& a, hipStream_t s);
struct ScatterArgs { RowMap map; int n; int val[kMaxRows]; int* dst; };   // dst[map.slot[i]] = val[i]
hipError_t launch_scatter_ints(const ScatterArgs& a, hipStream_t s);

// Memory-bound glue kernels of the path: rotary + KV-cache write, patch extraction (patch-embed as GEMM),
// embedding assembly for CLIP / pooler, the image-token splice gather, multi-view reductions, and the
// greedy-search sampling state machine (argmax with lowest-index tie-break, EOS / pad bookkeeping).
#include <stdio.h>

#include <mutex>

#include "../../include/oracle_hip.h"
#include "common.h"
#include "kernels.h"

// Scratch of orc_preprocess_images_noalloc (the uint8 image between the two resample passes): ONE array per GPU, held
// until the process ends. In host code this name is only a stand-in for the symbol: the per-device address comes from
// hipGetSymbolAddress (scratch_for_current_device) and only that pointer is ever handed to a kernel.
#define ORC_PREPROCESS_SCRATCH_BYTES (16u << 20)
static __device__ __align__(256) uint8_t scratch[ORC_PREPROCESS_SCRATCH_BYTES];

namespace {

// ------------------------------------------------------------------ rotary + KV write
// One workgroup per token row; a work unit = (head, 8-element chunk c of the low half) handles elements
// [8c, 8c+8) and their partners at +64 of q and k, plus the two v chunks.
// q' = bf16( bf16(q*cos) + bf16(rotate_half(q)*sin) ), rotate_half(x) = cat(-x[64:], x[:64]).
__global__ __launch_bounds__(256) void rope_kv_kernel(RopeArgs a) {
  const int row = blockIdx.x;
  const int b = row / a.T, t = row % a.T;
  const int sl = a.map.slot[b];                 // KV-cache slot of this batch row
  if (sl < 0) return;
  int pos, slot;
  if (a.pos) {
    pos = a.pos[row];
    slot = a.t0 + t;
  } else {
    const int len = a.cur_len[sl];
    pos = len - 1 - a.kbeg[sl];
    slot = len - 1;
  }
  pos = pos < 0 ? 0 : (pos >= a.maxpos ? a.maxpos - 1 : pos);
  const int D = a.H * 128;
  bf16_t* qr = a.qkv + (size_t)row * 3 * D;
  bf16_t* kr = qr + D;
  const bf16_t* vr = qr + 2 * D;
  for (int u = threadIdx.x; u < a.H * 8; u += 256) {
    const int head = u >> 3, c = u & 7;
    const u32x4 cs = *(const u32x4*)(a.cos_tab + (size_t)pos * 64 + c * 8);
    const u32x4 sn = *(const u32x4*)(a.sin_tab + (size_t)pos * 64 + c * 8);
    bf16_t* kdst = a.Kc + (size_t)sl * a.c_bs + (size_t)head * a.c_hs + (size_t)slot * 128;
    bf16_t* vdst = a.Vc + (size_t)sl * a.c_bs + (size_t)head * a.c_hs + (size_t)slot * 128;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      bf16_t* src = (which == 0 ? qr : kr) + head * 128 + c * 8;
      const u32x4 lo = *(const u32x4*)src;
      const u32x4 hi = *(const u32x4*)(src + 64);
      u32x4 olo, ohi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float c0 = lo_bf(cs[j]), c1 = hi_bf(cs[j]), s0 = lo_bf(sn[j]), s1 = hi_bf(sn[j]);
        const float l0 = lo_bf(lo[j]), l1 = hi_bf(lo[j]), h0 = lo_bf(hi[j]), h1 = hi_bf(hi[j]);
        olo[j] = pack2(rbf(l0 * c0) + rbf(-h0 * s0), rbf(l1 * c1) + rbf(-h1 * s1));
        ohi[j] = pack2(rbf(h0 * c0) + rbf(l0 * s0), rbf(h1 * c1) + rbf(l1 * s1));
      }
      if (which == 0) {
        *(u32x4*)src = olo;
        *(u32x4*)(src + 64) = ohi;
      } else {
        *(u32x4*)(kdst + c * 8) = olo;
        *(u32x4*)(kdst + 64 + c * 8) = ohi;
      }
    }
    *(u32x4*)(vdst + c * 8) = *(const u32x4*)(vr + head * 128 + c * 8);
    *(u32x4*)(vdst + 64 + c * 8) = *(const u32x4*)(vr + head * 128 + 64 + c * 8);
  }
}

// ------------------------------------------------------------------ patch extraction
// images (n,3,I,I) bf16 -> patches (n*G*G, kp), column k = c*P*P + ky*P + kx (Conv2d weight flattening),
// columns >= 3*P*P zero.
__global__ void im2col_kernel(const bf16_t* __restrict__ img, int n_img, int I, int P, bf16_t* __restrict__ out, int kp) {
  const int G = I / P;
  const size_t total = (size_t)n_img * G * G * kp;
  const int kreal = 3 * P * P;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(o % kp);
    const size_t pr = o / kp;
    bf16_t v = 0;
    if (k < kreal) {
      const int c = k / (P * P), ky = (k / P) % P, kx = k % P;
      const int px = (int)(pr % G), py = (int)((pr / G) % G);
      const size_t im = pr / ((size_t)G * G);
      v = img[((im * 3 + c) * I + (py * P + ky)) * I + px * P + kx];
    }
    out[o] = v;
  }
}

// e[img][t] = bf16( (t == 0 ? class_embedding : patch[img][t-1]) + position_embedding[t] )
__global__ __launch_bounds__(128) void clip_embed_kernel(const bf16_t* __restrict__ patches, const bf16_t* __restrict__ cls,
                                                         const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int tokens, int d) {
  const int row = blockIdx.x;
  const int img = row / tokens, t = row % tokens;
  const bf16_t* src = (t == 0) ? cls : patches + ((size_t)img * (tokens - 1) + (t - 1)) * d;
  for (int c = threadIdx.x; c < (d >> 3); c += 128) {
    const u32x4 x = *(const u32x4*)(src + c * 8);
    const u32x4 p = *(const u32x4*)(pos + (size_t)t * d + c * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2(lo_bf(x[j]) + lo_bf(p[j]), hi_bf(x[j]) + hi_bf(p[j]));
    *(u32x4*)(out + (size_t)row * d + c * 8) = o;
  }
}

// pooler input: e[b][s] = bf16( bf16(x + token_type[0]) + position[s] ), x = view features or 0 for padded views
__global__ __launch_bounds__(128) void bert_embed_kernel(const bf16_t* __restrict__ feats, const int* __restrict__ row_off,
                                                         const int* __restrict__ row_valid, const bf16_t* __restrict__ tt0,
                                                         const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int S, int d) {
  const int row = blockIdx.x;
  const int b = row / S, s = row % S;
  const bool valid = s < row_valid[b];
  const bf16_t* src = feats + ((size_t)row_off[b] + s) * d;
  for (int c = threadIdx.x; c < (d >> 3); c += 128) {
    u32x4 x = {0u, 0u, 0u, 0u};
    if (valid) x = *(const u32x4*)(src + c * 8);
    const u32x4 tt = *(const u32x4*)(tt0 + c * 8);
    const u32x4 p = *(const u32x4*)(pos + (size_t)s * d + c * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2(rbf(lo_bf(x[j]) + lo_bf(tt[j])) + lo_bf(p[j]), rbf(hi_bf(x[j]) + hi_bf(tt[j])) + hi_bf(p[j]));
    *(u32x4*)(out + (size_t)row * d + c * 8) = o;
  }
}

// drop the CLS row: out[img][t] = h[img][t+1]
__global__ __launch_bounds__(128) void strip_cls_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, int tokens, int d) {
  const int row = blockIdx.x;  // over n_img*(tokens-1)
  const int img = row / (tokens - 1), t = row % (tokens - 1);
  const bf16_t* src = h + ((size_t)img * tokens + t + 1) * d;
  for (int c = threadIdx.x; c < (d >> 3); c += 128) *(u32x4*)(out + (size_t)row * d + c * 8) = *(const u32x4*)(src + c * 8);
}

__global__ __launch_bounds__(128) void copy_rows_kernel(const bf16_t* __restrict__ src, long src_bs, bf16_t* __restrict__ dst,
                                                        long dst_bs, int rows, int d) {
  const int row = blockIdx.x;
  const int b = row / rows, r = row % rows;
  const bf16_t* s = src + (size_t)b * src_bs + (size_t)r * d;
  bf16_t* o = dst + (size_t)b * dst_bs + (size_t)r * d;
  for (int c = threadIdx.x; c < (d >> 3); c += 128) *(u32x4*)(o + c * 8) = *(const u32x4*)(s + c * 8);
}

// splice gather: code = kind<<28 | index; kind 0 zero row, 1 token embedding, 2 image feature row, 3 descriptor row
__global__ __launch_bounds__(128) void gather_rows_kernel(const int* __restrict__ codes, const bf16_t* __restrict__ tok,
                                                          const bf16_t* __restrict__ img, const bf16_t* __restrict__ desc,
                                                          bf16_t* __restrict__ out, int d) {
  const int row = blockIdx.x;
  const uint32_t code = (uint32_t)codes[row];
  const uint32_t kind = code >> 28, idx = code & 0x0FFFFFFFu;
  const bf16_t* src = kind == 1 ? tok : (kind == 2 ? img : desc);
  for (int c = threadIdx.x; c < (d >> 3); c += 128) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (kind != 0) v = *(const u32x4*)(src + (size_t)idx * d + c * 8);
    *(u32x4*)(out + (size_t)row * d + c * 8) = v;
  }
}

__global__ __launch_bounds__(128) void embed_tokens_kernel(const int* __restrict__ ids_by_row, const int* __restrict__ ids_by_slot,
                                                           RowMap map, const bf16_t* __restrict__ tok,
                                                           bf16_t* __restrict__ out, int d, int vocab,
                                                           float* __restrict__ ssq_part, int n_part, bf16_t* __restrict__ out_xf,
                                                           RowMetaArgs meta) {
  __shared__ float sh[2];
  const int row = blockIdx.x;
  const int sl = map.slot[row];
  if (meta.row_meta && threadIdx.x < 64) {   // this step's per-row operands of the q/k/v epilogues (gemv.hip RopePre)
    int len = 1, kb = 0, unf = 0, pos = 0;
    if (sl >= 0) {
      len = meta.cur_len[sl]; kb = meta.kbeg[sl]; unf = meta.unfinished ? meta.unfinished[sl] : 1;
      const int pos_raw = len - 1 - kb;
      pos = pos_raw < 0 ? 0 : (pos_raw >= meta.maxpos ? meta.maxpos - 1 : pos_raw);
    }
    if (threadIdx.x < 4) meta.row_meta[4 * row + threadIdx.x] = threadIdx.x == 0 ? sl : threadIdx.x == 1 ? len : threadIdx.x == 2 ? kb : unf;
    meta.row_cos[row * 64 + threadIdx.x] = meta.cos_tab[(size_t)pos * 64 + threadIdx.x];
    meta.row_sin[row * 64 + threadIdx.x] = meta.sin_tab[(size_t)pos * 64 + threadIdx.x];
  }
  int id = ids_by_row ? ids_by_row[row] : (sl < 0 ? 0 : ids_by_slot[sl]);
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  float ss = 0.f;
  for (int c = threadIdx.x; c < (d >> 3); c += 128) {
    const u32x4 v = *(const u32x4*)(tok + (size_t)id * d + c * 8);
    *(u32x4*)(out + (size_t)row * d + c * 8) = v;
    if (out_xf) *(u32x4*)(out_xf + xf_off_rows(row, c * 8, d)) = v;
#pragma unroll
    for (int j = 0; j < 4; ++j) ss += lo_bf(v[j]) * lo_bf(v[j]) + hi_bf(v[j]) * hi_bf(v[j]);
  }
  if (!ssq_part) return;
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
  __syncthreads();
  for (int t = threadIdx.x; t < n_part; t += 128) ssq_part[(size_t)row * n_part + t] = t == 0 ? sh[0] + sh[1] : 0.f;
}

// x (B, V, n) -> out (B, n): mode 0 = max over views, 1 = mean over views (fp32 sum, one rounding; torch mean)
__global__ void view_reduce_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int V, size_t n, int mode) {
  const size_t total = (size_t)B * n;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const size_t b = o / n, i = o % n;
    float acc = mode == 0 ? -INFINITY : 0.f;
    for (int v = 0; v < V; ++v) {
      const float f = bf2f(x[(b * V + v) * n + i]);
      acc = mode == 0 ? fmaxf(acc, f) : acc + f;
    }
    out[o] = f2bf(mode == 0 ? acc : acc / (float)V);
  }
}

// (n_img, grid*grid, d) -> (n_img, (grid/2)^2, d): 2x2 max pool over the token grid
__global__ void pool2x2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int n_img, int grid, int d) {
  const int g2 = grid / 2;
  const size_t total = (size_t)n_img * g2 * g2 * d;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(o % d);
    const size_t r = o / d;
    const int px = (int)(r % g2), py = (int)((r / g2) % g2);
    const size_t im = r / ((size_t)g2 * g2);
    float m = -INFINITY;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx)
        m = fmaxf(m, bf2f(x[((im * grid + (2 * py + dy)) * grid + (2 * px + dx)) * d + c]));
    out[o] = f2bf(m);
  }
}

// ------------------------------------------------------------------ image preprocessing
// expand2square + Pillow's 8-bit bicubic resize (two passes, uint8 in between, 22-bit fixed-point coefficient tables
// built on the host exactly as libImaging/Resample.c does) + /255 + mean/std normalisation (a 3 x 256 bf16 table).
// Integer arithmetic throughout: bit-exact with the host path (PIL + numpy float32) by construction.
// pass 1: rows of the padded square, horizontal taps.  grid (S, n), 128 threads over the output columns.
__global__ __launch_bounds__(128) void resize_h_kernel(const uint8_t* __restrict__ img, int H, int W, int S, int out,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef, int ksize,
                                                       int fr, int fg, int fb, uint8_t* __restrict__ tmp) {
  const int y = blockIdx.x, im = blockIdx.y;
  const int ox = (S - W) / 2, oy = (S - H) / 2;
  const int sy = y - oy;
  const bool row_in = sy >= 0 && sy < H;
  const uint8_t* src = img + ((size_t)im * H + (row_in ? sy : 0)) * W * 3;
  uint8_t* dst = tmp + ((size_t)im * S + y) * out * 3;
  for (int xx = threadIdx.x; xx < out; xx += 128) {
    const int x0 = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int t = 0; t < cnt; ++t) {
      const int k = coef[xx * ksize + t];
      const int sx = x0 + t - ox;
      int r = fr, g = fg, b = fb;
      if (row_in && sx >= 0 && sx < W) { r = src[sx * 3]; g = src[sx * 3 + 1]; b = src[sx * 3 + 2]; }
      a0 += r * k; a1 += g * k; a2 += b * k;
    }
    a0 >>= 22; a1 >>= 22; a2 >>= 22;
    dst[xx * 3] = (uint8_t)(a0 < 0 ? 0 : (a0 > 255 ? 255 : a0));
    dst[xx * 3 + 1] = (uint8_t)(a1 < 0 ? 0 : (a1 > 255 ? 255 : a1));
    dst[xx * 3 + 2] = (uint8_t)(a2 < 0 ? 0 : (a2 > 255 ? 255 : a2));
  }
}
// pass 2: vertical taps over the pass-1 image, then the normalisation table; output (n, 3, out, out) bf16.
__global__ __launch_bounds__(128) void resize_v_kernel(const uint8_t* __restrict__ tmp, int S, int out, const int* __restrict__ bounds,
                                                       const int* __restrict__ coef, int ksize, const bf16_t* __restrict__ lut,
                                                       bf16_t* __restrict__ dst) {
  const int yy = blockIdx.x, im = blockIdx.y;
  const int y0 = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const uint8_t* src = tmp + (size_t)im * S * out * 3;
  for (int xx = threadIdx.x; xx < out; xx += 128) {
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int t = 0; t < cnt; ++t) {
      const int k = coef[yy * ksize + t];
      const uint8_t* p = src + ((size_t)(y0 + t) * out + xx) * 3;
      a0 += p[0] * k; a1 += p[1] * k; a2 += p[2] * k;
    }
    a0 >>= 22; a1 >>= 22; a2 >>= 22;
    a0 = a0 < 0 ? 0 : (a0 > 255 ? 255 : a0);
    a1 = a1 < 0 ? 0 : (a1 > 255 ? 255 : a1);
    a2 = a2 < 0 ? 0 : (a2 > 255 ? 255 : a2);
    const size_t plane = (size_t)out * out;
    bf16_t* o = dst + (size_t)im * 3 * plane + (size_t)yy * out + xx;
    o[0] = lut[a0]; o[plane] = lut[256 + a1]; o[2 * plane] = lut[512 + a2];
  }
}

// ------------------------------------------------------------------ greedy sampling
// One workgroup per batch row: argmax over the vocabulary (lowest index on exact ties, as torch.argmax),
// then the greedy-search bookkeeping (transformers GenerationMixin greedy search):
//   nxt = unfinished ? argmax : pad;  unfinished &= nxt != eos.
// Block 0 of the LAST-arriving... no cross-block dependency is needed: per-row state only; the scalar
// counters are advanced by row 0's block after every row has been read by its own block (each block
// touches only its row, the counters are only read by LATER kernels of the step).
__global__ __launch_bounds__(1024) void argmax_sample_kernel(SampleArgs a) {   // 16 waves per row: 8 loads per thread at 32000
  __shared__ float sv[16];
  __shared__ int si[16];
  const int b = a.map.slot[blockIdx.x], tid = threadIdx.x, lane = tid & 63, w = tid >> 6;   // b = the slot whose state this row owns
  if (b < 0) return;
  const float* lg = a.logits + (size_t)blockIdx.x * a.vocab;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid * 4; i < a.vocab; i += 4096) {
    const f32x4 v = *(const f32x4*)(lg + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (v[j] > best || (v[j] == best && i + j < bi)) { best = v[j]; bi = i + j; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { sv[w] = best; si[w] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 16; ++i)
      if (sv[i] > best || (sv[i] == best && si[i] < bi)) { best = sv[i]; bi = si[i]; }
    if (bi == 0x7fffffff) bi = 0;   // all-NaN row: torch.argmax would return the first NaN; keep defined
    const int step = a.row_step[b];
    const int unf = a.unfinished[b];
    const int nxt = unf ? bi : a.pad_id;
    if (a.raw_argmax) a.raw_argmax[b] = bi;
    a.next_ids[b] = nxt;
    if (unf || !a.freeze_finished) {   // per-row state: only this block touches row b
      if (a.out_tokens && step < a.out_ld) a.out_tokens[(size_t)b * a.out_ld + step] = nxt;
      a.row_step[b] = step + 1;
      a.cur_len[b] += 1;
    }
    a.unfinished[b] = unf && (nxt != a.eos_id) && (step + 1 < a.limit[b]);
  }
}

// advances the device-side counters once per step (separate tiny kernel: runs after every row's block)
__global__ void step_advance_kernel(SampleArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int n = 0;
    for (int i = 0; i < a.n_count; ++i) {
      const int sl = a.count.slot[i];
      n += (sl >= 0 && a.unfinished[sl]) ? 1 : 0;
    }
    *a.n_unfinished = n;
    *a.step += 1;
  }
}

// unfinished = 1, kbeg, cur_len = T (keys cached after the prefill; the sampler advances it to T + 1 for the first decode
// step), step 0, the length stop; the slot's token row is filled with pad
__global__ __launch_bounds__(256) void slot_init_kernel(SlotInitArgs a) {
  const int i = blockIdx.x, sl = a.map.slot[i];
  if (sl < 0) return;
  if (threadIdx.x == 0) {
    a.unfinished[sl] = 1;
    a.kbeg_state[sl] = a.kbeg[i];
    a.cur_len[sl] = a.T;
    a.row_step[sl] = 0;
    a.limit_state[sl] = a.limit[i];
    a.pf_kbeg[i] = a.kbeg[i];
  }
  for (int t = threadIdx.x; t < a.out_ld; t += 256) a.tokens[(size_t)sl * a.out_ld + t] = a.pad_id;
}
__global__ void scatter_ints_kernel(ScatterArgs a) {
  const int i = threadIdx.x;
  if (i < a.n && a.map.slot[i] >= 0) a.dst[a.map.slot[i]] = a.val[i];
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b == 0 ? 1 : b));
}

}  // namespace

hipError_t launch_rope_kv(const RopeArgs& a, hipStream_t s) {
  if (a.B * a.T <= 0) return hipSuccess;
  hipLaunchKernelGGL(rope_kv_kernel, dim3(a.B * a.T), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_im2col(const bf16_t* img, int n_img, int image, int patch, bf16_t* out, int kp, hipStream_t s) {
  const int G = image / patch;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((size_t)n_img * G * G * kp)), dim3(256), 0, s, img, n_img, image, patch, out, kp);
  return hipGetLastError();
}
hipError_t launch_clip_embed(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, const bf16_t*, const bf16_t*,
                             bf16_t* out, int n_img, int tokens, int d, float, hipStream_t s) {
  hipLaunchKernelGGL(clip_embed_kernel, dim3(n_img * tokens), dim3(128), 0, s, patches, cls, pos, out, tokens, d);
  return hipGetLastError();
}
hipError_t launch_bert_embed(const bf16_t* feats, const int* row_off, const int* row_valid, const bf16_t* tt0,
                             const bf16_t* pos, const bf16_t*, const bf16_t*, bf16_t* out, int B, int S, int d, float,
                             hipStream_t s) {
  hipLaunchKernelGGL(bert_embed_kernel, dim3(B * S), dim3(128), 0, s, feats, row_off, row_valid, tt0, pos, out, S, d);
  return hipGetLastError();
}
hipError_t launch_strip_cls(const bf16_t* h, bf16_t* out, int n_img, int tokens, int d, hipStream_t s) {
  hipLaunchKernelGGL(strip_cls_kernel, dim3(n_img * (tokens - 1)), dim3(128), 0, s, h, out, tokens, d);
  return hipGetLastError();
}
hipError_t launch_copy_rows(const bf16_t* src, long src_bs, bf16_t* dst, long dst_bs, int B, int rows, int d, hipStream_t s) {
  if (B * rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(B * rows), dim3(128), 0, s, src, src_bs, dst, dst_bs, rows, d);
  return hipGetLastError();
}
hipError_t launch_gather_rows(const int* codes, const bf16_t* tok, const bf16_t* img, const bf16_t* desc, bf16_t* out,
                              int rows, int d, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(128), 0, s, codes, tok, img, desc, out, d);
  return hipGetLastError();
}
hipError_t launch_embed_tokens(const int* ids_by_row, const int* ids_by_slot, const RowMap& map, const bf16_t* tok, bf16_t* out,
                               int rows, int d, int vocab, float* ssq_part, int n_part, hipStream_t s, bf16_t* out_xf,
                               const RowMetaArgs* meta) {
  if (rows <= 0) return hipSuccess;
  if (rows > kMaxRows) return hipErrorInvalidValue;
  if (meta && meta->row_meta && !(meta->cur_len && meta->kbeg && meta->cos_tab && meta->sin_tab && meta->row_cos && meta->row_sin && meta->maxpos > 0))
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(rows), dim3(128), 0, s, ids_by_row, ids_by_slot, map, tok, out, d, vocab, ssq_part,
                     n_part, out_xf, meta ? *meta : RowMetaArgs());
  return hipGetLastError();
}
hipError_t launch_view_reduce(const bf16_t* x, bf16_t* out, int B, int V, int n, int mode, hipStream_t s) {
  hipLaunchKernelGGL(view_reduce_kernel, dim3(grid_for((size_t)B * n)), dim3(256), 0, s, x, out, B, V, (size_t)n, mode);
  return hipGetLastError();
}
hipError_t launch_pool2x2(const bf16_t* x, bf16_t* out, int n_img, int grid, int d, hipStream_t s) {
  hipLaunchKernelGGL(pool2x2_kernel, dim3(grid_for((size_t)n_img * (grid / 2) * (grid / 2) * d)), dim3(256), 0, s, x, out, n_img, grid, d);
  return hipGetLastError();
}
hipError_t launch_preprocess(const uint8_t* img, int n, int H, int W, int out, const int* bounds, const int* coef, int ksize,
                             const bf16_t* lut, int fr, int fg, int fb, uint8_t* tmp, bf16_t* dst, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (H <= 0 || W <= 0 || out <= 0 || ksize <= 0) return hipErrorInvalidValue;
  const int S = H > W ? H : W;
  hipLaunchKernelGGL(resize_h_kernel, dim3(S, n), dim3(128), 0, s, img, H, W, S, out, bounds, coef, ksize, fr, fg, fb, tmp);
  hipLaunchKernelGGL(resize_v_kernel, dim3(out, n), dim3(128), 0, s, tmp, S, out, bounds, coef, ksize, lut, dst);
  return hipGetLastError();
}
hipError_t launch_slot_init(const SlotInitArgs& a, hipStream_t s) {
  if (a.k < 1 || a.k > kMaxRows) return hipErrorInvalidValue;
  hipLaunchKernelGGL(slot_init_kernel, dim3(a.k), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_scatter_ints(const ScatterArgs& a, hipStream_t s) {
  if (a.n < 1 || a.n > kMaxRows) return hipErrorInvalidValue;
  hipLaunchKernelGGL(scatter_ints_kernel, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_argmax_sample(const SampleArgs& a, hipStream_t s) {
  if ((a.vocab & 3) || a.B <= 0 || a.B > kMaxRows || a.n_count > kMaxRows) return hipErrorInvalidValue;
  hipLaunchKernelGGL(argmax_sample_kernel, dim3(a.B), dim3(1024), 0, s, a);
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ allocation-free preprocessing entry point
namespace {
constexpr int kMaxDevices = 64;
uint8_t* g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each GPU, resolved once per device
std::mutex g_scratch_mu;

// the current device's copy of `scratch` (each GPU holds its own array at its own address)
int scratch_for_current_device(uint8_t** out) {
  int dev = -1;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return orc_fail("hipGetDevice", hipGetErrorString(err));
  if (dev < 0 || dev >= kMaxDevices) return orc_fail("orc_preprocess_images_noalloc", "device index out of range");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (err != hipSuccess || !p) return orc_fail("hipGetSymbolAddress(scratch)", hipGetErrorString(err));
    g_scratch_ptr[dev] = (uint8_t*)p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

extern "C" int64_t orc_preprocess_noalloc_capacity(void) { return (int64_t)ORC_PREPROCESS_SCRATCH_BYTES; }

extern "C" int orc_preprocess_images_noalloc(const uint8_t* rgb_hwc, int n_images, int height, int width, int out_size,
                                             const int32_t* bounds, const int32_t* coeffs, int ksize, const void* lut_bf16,
                                             const int32_t* fill_rgb_host, void* out, void* stream) {
  uint8_t* tmp = nullptr;
  if (int rc = scratch_for_current_device(&tmp)) return rc;
  if (n_images == 0) return 0;   // set-up call: resolves this device's scratch address (do it once outside graph capture)
  if (!rgb_hwc || !bounds || !coeffs || !lut_bf16 || !fill_rgb_host || !out) return orc_fail("orc_preprocess_images_noalloc", "null argument");
  if (n_images < 0 || height < 1 || width < 1 || out_size < 1 || ksize < 1) return orc_fail("orc_preprocess_images_noalloc", "bad size");
  const int64_t side = height > width ? height : width;
  const int64_t need = (int64_t)n_images * side * out_size * 3;
  if (need > (int64_t)ORC_PREPROCESS_SCRATCH_BYTES) {
    char msg[200];
    snprintf(msg, sizeof(msg), "needs %lld bytes of scratch (n_images * max(height, width) * out_size * 3), the static buffer holds %u",
             (long long)need, ORC_PREPROCESS_SCRATCH_BYTES);
    return orc_fail("orc_preprocess_images_noalloc", msg);
  }
  // pass 1 writes every byte of tmp[0, need) before pass 2 reads it (same stream): nothing of an earlier call is read
  hipError_t err = launch_preprocess(rgb_hwc, n_images, height, width, out_size, bounds, coeffs, ksize, (const bf16_t*)lut_bf16,
                                     fill_rgb_host[0], fill_rgb_host[1], fill_rgb_host[2], tmp, (bf16_t*)out, (hipStream_t)stream);
  if (err != hipSuccess) return orc_fail("orc_preprocess_images_noalloc", hipGetErrorString(err));
  return 0;
}

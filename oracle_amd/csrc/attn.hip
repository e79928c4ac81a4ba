// Flash-style attention on v_mfma_f32_16x16x32_bf16 for the three dense attention shapes of the path:
//   CLIP ViT (16 heads x 64, 577 tokens, no mask), pooler BERT (8 heads x 128, key-padding mask),
//   LLaMA prefill (32 heads x 128, causal + left-pad mask).
// and a split-KV streaming kernel for decode (one query per row, HBM-bound on the KV cache).
//
// Prefill kernel: one workgroup = 4 waves = 128 query rows, each wave 2 x 16 rows; K/V tiles of 64 keys staged
// in LDS by LDS-DMA (XOR-swizzled so both the ds_read_b128 K reads and the ds_read_b64_tr_b16 V reads are
// conflict-free). Orientation is chosen so that no data crosses lanes between the two products:
//   S^T = K . Q^T   (A = K tile from LDS, B = Q^T fragment held in registers)  -> lane holds, for its
//                    query column q = lane&15, keys 4g..4g+3 of each 16-key block (g = lane>>4)
//   O^T = V^T . P^T (A = V^T via the transposing LDS read, B = P^T = the lane's own probabilities)
// so row max / row sum / rescale are per-lane scalars (+2 shuffles across g), and the k-slot order of the
// second product is simply defined as "the order the first product left the keys in".
// Rounding points follow the reference's eager path: scores rounded to bf16 after QK^T and again after
// the scale multiply; probabilities rounded to bf16 before PV (here before the final 1/l).
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int HD> __device__ __forceinline__ int fK(int key) { return HD == 128 ? (key & 15) : ((key >> 1) & 7); }
template <int HD> __device__ __forceinline__ int fV(int key) { return HD == 128 ? (key & 7) : ((key >> 1) & 3); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// One workgroup = 4 waves = 128 query rows (each wave two 16-row blocks, so every K / V fragment read from LDS feeds two
// MFMAs); K/V tiles of 64 keys arrive by LDS-DMA (global_load_lds, 16 B per lane) into a 2-deep ring:
//     wait my DMA of stage s -> barrier -> issue the DMA of tile t+1 -> compute on tile t
// The XOR swizzles of the K image (ds_read_b128) and of the V image (ds_read_b64_tr_b16) permute 16-B chunks inside a
// row, so they are applied to the per-lane SOURCE address (LDS-DMA destinations are lane-linear).
// QB = 16-row query blocks per wave. QB = 2 (4 waves, 204 VGPRs at head dim 128: 2 waves per SIMD): every K / V fragment read
// from LDS feeds two MFMAs. QB = 1 (r3: 8 waves of one query block, 124 VGPRs: 4 waves per SIMD): twice the LDS reads per MFMA but
// twice the waves to hide a wave's serial QK^T -> softmax -> PV chain. Same arithmetic per query row either way (same bits).
// Measured in the bench (kernel trace): head dim 128 (prefill, pooler) 87.9 -> 82.1 us with QB = 1, head dim 64 (ViT) 94.3 -> 96.4:
// the default is QB = 1 at 128 and QB = 2 at 64; ORC_ATTN_QB=1 / 2 forces one form for A/B runs. Work-wise the prefill launch is
// ~14 us of MFMA, ~28 us of softmax VALU and 17.5 (QB = 2) / 35 us (QB = 1) of LDS reads: both forms are bound by the latency of the
// dependent chain at few waves per SIMD, not by a pipe (the two-tile software pipeline stays undone: DESIGN 9). Neither more
// independent workgroups per CU (64-row workgroups of 4 waves on 32-key tiles, four per CU: 88 -> 101 us; ViT five per CU: 96 -> 99),
// nor 32-key tiles at 8 waves (96 us), nor the ring stage as a compile-time constant (~25 fewer VALU per tile: 89.3 -> 89.3 us) moved
// it, nor did v_permlane16/32_swap in place of the four ds_bpermute of the softmax reductions (bit-equal, 90.6 -> 90.4 us):
// profiles/r3/attn_bench_variants.log, tools/experiments/r3_attn_*.patch. The counters there say why: VALU issue is 62 % of the launch
// and MFMA 26 %, almost serially -- the ~1100 VALU cycles of softmax per wave and tile (two bf16 roundings per score as the
// reference makes them, 17 v_exp) are the floor of this formulation, LDS conflicts are zero.
template <int HD, int QB>
__global__ __launch_bounds__(QB == 2 ? 256 : 512, QB == 2 ? 2 : 4) void attn_kernel(AttnArgs a, int order) {
  constexpr int KB = 64;
  constexpr int NTHR = QB == 2 ? 256 : 512;
  constexpr int WROWS = 16 * QB;             // query rows per wave
  constexpr int ROWB = HD * 2;
  constexpr int CH = HD / 8;                 // 16-byte chunks per key row
  constexpr int TILE = KB * ROWB;            // bytes of one K (or V) tile
  constexpr int RND = KB * CH / NTHR;        // DMA instructions per wave per operand per tile
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [stage][K | V]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  // Workgroup -> (query block, head, row). Consecutive workgroup ids go round-robin over the 8 XCDs, so with
  // (head, row) pairs dealt to XCDs the query blocks that share one K/V run on the same XCD (one L2 fill), and the
  // causal case issues the long blocks (most visible keys) first.
  int b, h, qt;
  {
    const int nq = (a.Sq + 127) / 128;
    const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
    if (order == 1) {
      const int pair = (slot / nq) * 8 + xcd;
      qt = nq - 1 - slot % nq;
      h = pair % a.H; b = pair / a.H;
    } else if (order == 2) {
      const int per = (a.H * a.B) >> 3;
      const int pair = (slot % per) * 8 + xcd;
      qt = nq - 1 - slot / per;
      h = pair % a.H; b = pair / a.H;
    } else {
      qt = lin % nq; h = (lin / nq) % a.H; b = lin / (nq * a.H);
    }
  }
  const int qblk = qt * 128;
  const int kb_ = a.kbeg ? a.kbeg[b] : 0;
  const int ke_ = a.kend ? a.kend[b] : a.Sk;

  bf16x8 qf[QB][HD / 32];
  int qi[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = qblk + w * WROWS + qb * 16 + r16;
    const int qrow_c = qrow < a.Sq ? qrow : a.Sq - 1;
    const bf16_t* qp = a.Q + (size_t)b * a.q_bs + (size_t)h * a.q_hs + (size_t)qrow_c * a.q_ts;
#pragma unroll
    for (int kk = 0; kk < HD / 32; ++kk) qf[qb][kk] = *(const bf16x8*)(qp + kk * 32 + g * 8);
    qi[qb] = qrow + a.q_off;
  }
  int q_hi = qblk + 127;
  if (q_hi > a.Sq - 1) q_hi = a.Sq - 1;
  int k_hi = ke_;
  if (a.causal && q_hi + a.q_off + 1 < k_hi) k_hi = q_hi + a.q_off + 1;
  const int qw_hi = qblk + w * WROWS + WROWS - 1 + a.q_off;            // last key this wave can see under the causal mask

  f32x4 o[QB][HD / 16];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int i = 0; i < HD / 16; ++i) o[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[QB], l[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { m[qb] = -INFINITY; l[qb] = 0.f; }

  const int kvb = a.kv_map_on ? a.kv_map.slot[b] : b;        // KV-cache slot of this batch row (prefill into arbitrary slots)
  const bf16_t* kbase = a.K + (size_t)kvb * a.k_bs + (size_t)h * a.k_hs;
  const bf16_t* vbase = a.V + (size_t)kvb * a.v_bs + (size_t)h * a.v_hs;
  // DMA addressing: lane p of round r fills 16-B slot r*256 + p of the image, i.e. key r*(256/CH) + p/CH, chunk p%CH.
  // The per-lane part (key row inside the round, swizzled source chunk) is loop-invariant and the per-round / per-tile
  // part is wave-uniform, so a tile costs no vector address arithmetic; only the tile that crosses Sk clamps per lane.
  constexpr int KPR = NTHR / CH;              // keys per DMA round
  const int p_ = w * 64 + lane;
  const int key_p = p_ / CH, cp_ = p_ % CH;
  int ck_r[RND], cv_r[RND];                  // swizzled source chunk (depends on the key, so on r) -- folds to constants
  unsigned koff[RND], voff[RND];
#pragma unroll
  for (int r = 0; r < RND; ++r) {
    const int key = r * KPR + key_p;
    ck_r[r] = cp_ ^ fK<HD>(key);
    cv_r[r] = (((cp_ >> 1) ^ fV<HD>(key)) << 1) | (cp_ & 1);
    koff[r] = (unsigned)(key_p * a.k_ts + ck_r[r] * 8);
    voff[r] = (unsigned)(key_p * a.v_ts + cv_r[r] * 8);
  }
  auto dma = [&](int k0, int stage) {
    char* sk = smem + stage * 2 * TILE;
    char* sv = sk + TILE;
    if (k0 + KB <= a.Sk) {
#pragma unroll
      for (int r = 0; r < RND; ++r) {
        const bf16_t* kr = kbase + (size_t)(k0 + r * KPR) * a.k_ts;       // wave-uniform
        const bf16_t* vr = vbase + (size_t)(k0 + r * KPR) * a.v_ts;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(kr + koff[r]), (lds_void_t*)(sk + (r * NTHR + w * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(vr + voff[r]), (lds_void_t*)(sv + (r * NTHR + w * 64) * 16), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < RND; ++r) {
        int gk = k0 + r * KPR + key_p;
        gk = gk < a.Sk ? gk : a.Sk - 1;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(kbase + (size_t)gk * a.k_ts + ck_r[r] * 8), (lds_void_t*)(sk + (r * NTHR + w * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(vbase + (size_t)gk * a.v_ts + cv_r[r] * 8), (lds_void_t*)(sv + (r * NTHR + w * 64) * 16), 16, 0, 0);
      }
    }
  };

  const int k_first = (kb_ / KB) * KB;
  if (k_first < k_hi) dma(k_first, 0);
  int stage = 0;
  for (int k0 = k_first; k0 < k_hi; k0 += KB, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k0 + KB < k_hi) dma(k0 + KB, stage ^ 1);
    if (a.causal && k0 > qw_hi) continue;                    // every key of this tile is in this wave's future
    const char* sK = smem + stage * 2 * TILE;
    const char* sV = sK + TILE;

    // scores: bf16 rounding after QK^T and again after the scale (the reference's two eager ops). The softmax below is
    // VALU-bound (about 10 vector ops per score against 1/8 of an MFMA), so a tile that no mask touches -- every tile but
    // the padding edge and the causal diagonal -- skips the per-element key tests.
    const bool need_mask = k0 < kb_ || k0 + KB > ke_ || (a.causal && k0 + KB - 1 > qblk + w * WROWS + a.q_off);
    float s[QB][16];
#pragma unroll
    for (int kb4 = 0; kb4 < 4; ++kb4) {
      f32x4 acc[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) acc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int key_l = kb4 * 16 + r16;
#pragma unroll
      for (int kk = 0; kk < HD / 32; ++kk) {
        const int c = kk * 4 + g;
        const bf16x8 kf = *(const bf16x8*)(sK + key_l * ROWB + ((c ^ fK<HD>(key_l)) << 4));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][kk], acc[qb], 0, 0, 0);
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          float x0, x1;
          rbf2(acc[qb][r], acc[qb][r + 1], x0, x1);
          rbf2(x0 * a.scale, x1 * a.scale, x0, x1);
          s[qb][kb4 * 4 + r] = x0;
          s[qb][kb4 * 4 + r + 1] = x1;
        }
    }
    if (need_mask) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key_abs = k0 + (j >> 2) * 16 + 4 * g + (j & 3);
          const bool ok = key_abs >= kb_ && key_abs < ke_ && (!a.causal || key_abs <= qi[qb]);
          s[qb][j] = ok ? s[qb][j] : -INFINITY;
        }
    }
    bf16x8 pf[QB][2];
    float alpha[QB];
    constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float tm = s[qb][0];
#pragma unroll
      for (int j = 1; j < 16; ++j) tm = fmaxf(tm, s[qb][j]);
      tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float m_new = fmaxf(m[qb], tm);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float mb = -m_use * LOG2E;
      alpha[qb] = __builtin_amdgcn_exp2f(fmaf(m[qb], LOG2E, mb));     // exp2(-inf) = 0 covers the first tile
      float rs = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[qb][j], LOG2E, mb));   // masked keys: exp2(-inf) = 0
        rs += p;
        pf[qb][j >> 3][j & 7] = (__bf16)p;
      }
      rs += __shfl_xor(rs, 16, 64);
      rs += __shfl_xor(rs, 32, 64);
      l[qb] = l[qb] * alpha[qb] + rs;
      m[qb] = m_new;
    }
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[qb][dt] *= alpha[qb];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int key_a = hh * 32 + 4 * g + (r16 >> 2);
        const int within = 8 * (r16 & 3);
        const char* p0 = sV + key_a * ROWB + ((dt ^ fV<HD>(key_a)) << 5) + within;
        const char* p1 = sV + (key_a + 16) * ROWB + ((dt ^ fV<HD>(key_a + 16)) << 5) + within;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p1);
        const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][hh], o[qb][dt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = qblk + w * WROWS + qb * 16 + r16;
    if (qrow >= a.Sq) continue;
    const float inv = l[qb] > 0.f ? 1.0f / l[qb] : 0.f;
    bf16_t* op = a.O + (size_t)b * a.o_bs + (size_t)h * a.o_hs + (size_t)qrow * a.o_ts;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      u32x2 st = {pack2(o[qb][dt][0] * inv, o[qb][dt][1] * inv), pack2(o[qb][dt][2] * inv, o[qb][dt][3] * inv)};
      *(u32x2*)(op + dt * 16 + 4 * g) = st;
    }
  }
}

// ------------------------------------------------------------------------------------------ decode
__device__ __forceinline__ float rescale(float m_old, float m_new) {
  return (m_old == -INFINITY) ? 0.f : __expf(m_old - m_new);
}

// grid (H, B, nsplit), WAVES*64 threads. Lane layout: 16 lanes per key (8 head-dim elements each),
// 4 keys per wave-instruction = one contiguous 1 KiB read of the cache. With nsplit == 1 (enough (row, head) pairs to
// fill the chip) the workgroup owns the whole key range and writes the bf16 output itself; otherwise partial
// (m, l, acc) go to the workspace and decode_attn_combine_kernel merges them.
template <int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void decode_attn_kernel(DecodeAttnArgs a) {
  __shared__ float sh_m[WAVES], sh_l[WAVES];
  __shared__ __attribute__((aligned(16))) float sh_acc[WAVES][128];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, c = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z, nsplit = gridDim.z;
  const int sl = a.map.slot[b];
  const int len = sl < 0 ? 1 : a.cur_len[sl];
  const int kb = sl < 0 ? 0 : a.kbeg[sl];
  if (sl < 0 || (a.unfinished && !a.unfinished[sl])) {   // inactive or finished row (its tokens are pad): no cache read at all
    if (tid < 128) {
      if (nsplit == 1) {
        a.O[a.o_xf ? xf_off_rows(b, h * 128 + tid, a.H * 128) : (size_t)b * a.o_bs + h * 128 + tid] = 0;
      } else {
        float* pp = a.part + (((size_t)b * a.H + h) * DECODE_NSPLIT + sp) * 130;
        if (tid == 0) { pp[0] = -INFINITY; pp[1] = 0.f; }
        pp[2 + tid] = 0.f;
      }
    }
    return;
  }
  const int n = len - kb;
  const int per = (n + nsplit - 1) / nsplit;
  const int k_lo = kb + sp * per;
  const int k_hi = (k_lo + per < len) ? k_lo + per : len;

  float q[8];
  {
    const u32x4 qv = *(const u32x4*)(a.Q + (size_t)b * a.q_bs + h * 128 + c * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) { q[2 * j] = lo_bf(qv[j]); q[2 * j + 1] = hi_bf(qv[j]); }
  }
  const bf16_t* kc = a.Kc + (size_t)sl * a.c_bs + (size_t)h * a.c_hs + c * 8;
  const bf16_t* vc = a.Vc + (size_t)sl * a.c_bs + (size_t)h * a.c_hs + c * 8;

  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;

  for (int base = k_lo + w * 4; base < k_hi; base += 4 * WAVES * U) {
    u32x4 kv[U], vv[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = base + u * 4 * WAVES + kq;
      ok[u] = key < k_hi;
      const int kc_ = ok[u] ? key : (k_hi - 1 > 0 ? k_hi - 1 : 0);
      kv[u] = __builtin_nontemporal_load((const u32x4*)(kc + (size_t)kc_ * 128));
      vv[u] = __builtin_nontemporal_load((const u32x4*)(vc + (size_t)kc_ * 128));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) d += q[2 * j] * lo_bf(kv[u][j]) + q[2 * j + 1] * hi_bf(kv[u][j]);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      d += __shfl_xor(d, 8, 64);
      const float s = ok[u] ? rbf(rbf(d) * a.scale) : -INFINITY;
      const float m_new = fmaxf(m, s);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = rescale(m, m_use);
      const float p = (s == -INFINITY) ? 0.f : __expf(s - m_use);
      const float pb = rbf(p);
      l = l * alpha + p;
      m = m_new;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] = acc[2 * j] * alpha + pb * lo_bf(vv[u][j]);
        acc[2 * j + 1] = acc[2 * j + 1] * alpha + pb * hi_bf(vv[u][j]);
      }
    }
  }
  // merge the 4 key slots of the wave
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m_o = __shfl_xor(m, off, 64);
    const float l_o = __shfl_xor(l, off, 64);
    const float m_n = fmaxf(m, m_o);
    const float m_use = (m_n == -INFINITY) ? 0.f : m_n;
    const float a0 = rescale(m, m_use), a1 = rescale(m_o, m_use);
    l = l * a0 + l_o * a1;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * a0 + __shfl_xor(acc[j], off, 64) * a1;
    m = m_n;
  }
  if (kq == 0) {
    if (c == 0) { sh_m[w] = m; sh_l[w] = l; }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh_acc[w][c * 8 + j] = acc[j];
  }
  __syncthreads();
  if (tid < 128) {
    float mm = -INFINITY;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) mm = fmaxf(mm, sh_m[i]);
    const float m_use = (mm == -INFINITY) ? 0.f : mm;
    float ll = 0.f, av = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
      const float sc = rescale(sh_m[i], m_use);
      ll += sh_l[i] * sc;
      av += sh_acc[i][tid] * sc;
    }
    if (nsplit == 1) {
      a.O[a.o_xf ? xf_off_rows(b, h * 128 + tid, a.H * 128) : (size_t)b * a.o_bs + h * 128 + tid] = f2bf(ll > 0.f ? av / ll : 0.f);
    } else {
      float* pp = a.part + (((size_t)b * a.H + h) * DECODE_NSPLIT + sp) * 130;
      if (tid == 0) { pp[0] = mm; pp[1] = ll; }
      pp[2 + tid] = av;
    }
  }
}

// ---- r3: the same lane layout and the same rounding points, as a software pipeline. decode_attn_kernel above issues a step's
// K and V loads, waits, computes, and only then requests the next step (a wave has nothing in flight while it runs its
// dependent dot -> 4 LDS-crossbar shuffles (each waited for) -> exp -> accumulate chain: .s of r2). Here:
//   * K / V rows come through buffer descriptors whose range ends at the row's last cached key: the load of step i + 1 is issued
//     unconditionally BEFORE step i is computed (two named register sets, counted vmcnt), keys past the end return zeros
//     without a memory request and without a branch;
//   * the 16-lane dot-product reduction runs on DPP row operations (4 v_add_f32_dpp, no LDS traffic, no lgkmcnt wait) and the
//     products on v_dot2c_f32_bf16 (q stays packed bf16).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, in every lane of the row
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}
// (by value on purpose: __builtin_bit_cast applied directly to a vector-element expression read element 0 for every index -- hipcc 7.2)
__device__ __forceinline__ bf16x2_t as_bf2(uint32_t x) { return __builtin_bit_cast(bf16x2_t, x); }
template <int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void decode_attn_pipe_kernel(DecodeAttnArgs a) {
  __shared__ float sh_m[WAVES], sh_l[WAVES];
  __shared__ __attribute__((aligned(16))) float sh_acc[WAVES][128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, c = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z, nsplit = gridDim.z;
#ifdef GEMV_ALL_VARIANTS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  if (a.dbg_ts) ts0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int sl = a.map.slot[b];
  const int len = sl < 0 ? 1 : a.cur_len[sl];
  const int kb = sl < 0 ? 0 : a.kbeg[sl];
#ifdef GEMV_ALL_VARIANTS
  if (a.dbg_ts) { asm volatile("" :: "s"(len), "s"(kb)); ts1 = __builtin_amdgcn_s_memrealtime(); }
#endif
  if (sl < 0 || (a.unfinished && !a.unfinished[sl])) {   // inactive or finished row (its tokens are pad): no cache read at all
    if (tid < 128) {
      if (nsplit == 1) {
        a.O[a.o_xf ? xf_off_rows(b, h * 128 + tid, a.H * 128) : (size_t)b * a.o_bs + h * 128 + tid] = 0;
      } else {
        float* pp = a.part + (((size_t)b * a.H + h) * DECODE_NSPLIT + sp) * 130;
        if (tid == 0) { pp[0] = -INFINITY; pp[1] = 0.f; }
        pp[2 + tid] = 0.f;
      }
    }
    return;
  }
  const int n = len - kb;
  const int per = (n + nsplit - 1) / nsplit;
  const int k_lo = kb + sp * per;
  const int k_hi = (k_lo + per < len) ? k_lo + per : len;

  const u32x4 qv = *(const u32x4*)(a.Q + (size_t)b * a.q_bs + h * 128 + c * 8);
  // descriptors over this (slot, head)'s rows [0, k_hi): a key index >= k_hi is out of range
  const bf16_t* kbase = a.Kc + (size_t)sl * a.c_bs + (size_t)h * a.c_hs;
  const bf16_t* vbase = a.Vc + (size_t)sl * a.c_bs + (size_t)h * a.c_hs;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, k_hi * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, k_hi * 256, 0x00020000);
  const int loff = kq * 256 + c * 16;                                 // this lane's bytes inside a 4-key step
  constexpr int STEP = 4 * WAVES;                                     // keys per workgroup step

  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;

  // the whole offset travels in the per-lane register (the range check then needs nothing from the scalar offset)
  auto load = [&](u32x4 (&kd)[U], u32x4 (&vd)[U], int base) {        // keys base + u * STEP + kq (base wave-uniform)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int off = loff + (base + u * STEP) * 256;
      kd[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 2));
      vd[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 2));
    }
  };
  auto step = [&](const u32x4 (&kd)[U], const u32x4 (&vd)[U], int base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_fdot2_f32_bf16(as_bf2(qv[j]), as_bf2(kd[u][j]), d, false);
      d = row16_sum(d);
      const bool ok = base + u * STEP + kq < k_hi;
      const float sc = ok ? rbf(rbf(d) * a.scale) : -INFINITY;
      const float m_new = fmaxf(m, sc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = rescale(m, m_use);
      const float p = (sc == -INFINITY) ? 0.f : __expf(sc - m_use);
      const float pb = rbf(p);
      l = l * alpha + p;
      m = m_new;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] = acc[2 * j] * alpha + pb * lo_bf(vd[u][j]);
        acc[2 * j + 1] = acc[2 * j + 1] * alpha + pb * hi_bf(vd[u][j]);
      }
    }
  };
  {
    u32x4 kA[U], vA[U], kB[U], vB[U];
    int base = k_lo + w * 4;
    load(kA, vA, base);
    while (base < k_hi) {
      load(kB, vB, base + U * STEP);
      step(kA, vA, base);
      base += U * STEP;
      if (base >= k_hi) break;
      load(kA, vA, base + U * STEP);
      step(kB, vB, base);
      base += U * STEP;
    }
  }
#ifdef GEMV_ALL_VARIANTS
  if (a.dbg_ts) {
    ts2 = __builtin_amdgcn_s_memrealtime();
    const size_t nb = (size_t)gridDim.x * gridDim.y, blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (lane == 0) a.dbg_ts[nb * 4 + blk * 16 + w] = ts2 - ts0;
  }
#endif
  // merge the 4 key slots of the wave
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m_o = __shfl_xor(m, off, 64);
    const float l_o = __shfl_xor(l, off, 64);
    const float m_n = fmaxf(m, m_o);
    const float m_use = (m_n == -INFINITY) ? 0.f : m_n;
    const float a0 = rescale(m, m_use), a1 = rescale(m_o, m_use);
    l = l * a0 + l_o * a1;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * a0 + __shfl_xor(acc[j], off, 64) * a1;
    m = m_n;
  }
  if (kq == 0) {
    if (c == 0) { sh_m[w] = m; sh_l[w] = l; }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh_acc[w][c * 8 + j] = acc[j];
  }
  __syncthreads();
  if (tid < 128) {
    float mm = -INFINITY;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) mm = fmaxf(mm, sh_m[i]);
    const float m_use = (mm == -INFINITY) ? 0.f : mm;
    float ll = 0.f, av = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
      const float sc = rescale(sh_m[i], m_use);
      ll += sh_l[i] * sc;
      av += sh_acc[i][tid] * sc;
    }
    if (nsplit == 1) {
      a.O[a.o_xf ? xf_off_rows(b, h * 128 + tid, a.H * 128) : (size_t)b * a.o_bs + h * 128 + tid] = f2bf(ll > 0.f ? av / ll : 0.f);
    } else {
      float* pp = a.part + (((size_t)b * a.H + h) * DECODE_NSPLIT + sp) * 130;
      if (tid == 0) { pp[0] = mm; pp[1] = ll; }
      pp[2 + tid] = av;
    }
  }
#ifdef GEMV_ALL_VARIANTS
  if (a.dbg_ts && w == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* o = a.dbg_ts + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
    }
  }
#endif
}

__global__ __launch_bounds__(128) void decode_attn_combine_kernel(DecodeAttnArgs a, int nsplit) {
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* pp = a.part + ((size_t)b * a.H + h) * DECODE_NSPLIT * 130;
  float mm = -INFINITY;
  for (int i = 0; i < nsplit; ++i) mm = fmaxf(mm, pp[i * 130]);
  const float m_use = (mm == -INFINITY) ? 0.f : mm;
  float ll = 0.f, av = 0.f;
  for (int i = 0; i < nsplit; ++i) {
    const float sc = rescale(pp[i * 130], m_use);
    ll += pp[i * 130 + 1] * sc;
    av += pp[i * 130 + 2 + tid] * sc;
  }
  a.O[a.o_xf ? xf_off_rows(b, h * 128 + tid, a.H * 128) : (size_t)b * a.o_bs + h * 128 + tid] = f2bf(ll > 0.f ? av / ll : 0.f);
}

}  // namespace

hipError_t launch_attention(const AttnArgs& a, int head_dim, hipStream_t s) {
  if (a.B <= 0 || a.Sq <= 0 || a.Sk <= 0) return hipSuccess;
  dim3 grid((unsigned)((a.Sq + 127) / 128) * a.H * a.B);
  // causal: longest query blocks first over the whole grid (measured 111 -> 92 us at 8 x 32 heads x 739 tokens, the
  // tail of late long blocks was 17 % of the kernel); otherwise the query blocks of one (head, row) share an XCD.
  const int order = ((a.H * a.B) & 7) ? 0 : (a.causal ? 2 : 1);
  static const int qb_env = []() { const char* e = getenv("ORC_ATTN_QB"); return e && (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0; }();
  const int qb = qb_env ? qb_env : (head_dim == 128 ? 1 : 2);
  if (head_dim == 128 && qb == 2)
    hipLaunchKernelGGL((attn_kernel<128, 2>), grid, dim3(256), 0, s, a, order);
  else if (head_dim == 128)
    hipLaunchKernelGGL((attn_kernel<128, 1>), grid, dim3(512), 0, s, a, order);
  else if (head_dim == 64 && qb == 2)
    hipLaunchKernelGGL((attn_kernel<64, 2>), grid, dim3(256), 0, s, a, order);
  else if (head_dim == 64)
    hipLaunchKernelGGL((attn_kernel<64, 1>), grid, dim3(512), 0, s, a, order);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_decode_attention(const DecodeAttnArgs& a, hipStream_t s, int variant) {
  if (a.B <= 0) return hipSuccess;
  if (a.B * a.H >= 192) {   // one workgroup per (row, head) fills the chip without a KV split
    switch (variant) {
      case 100: hipLaunchKernelGGL((decode_attn_kernel<16, 1>), dim3(a.H, a.B, 1), dim3(1024), 0, s, a); break;   // r2 kernel (tests / tools)
#ifdef GEMV_ALL_VARIANTS
      case 12: hipLaunchKernelGGL((decode_attn_pipe_kernel<16, 1>), dim3(a.H, a.B, 1), dim3(1024), 0, s, a); break;
      case 13: hipLaunchKernelGGL((decode_attn_pipe_kernel<16, 2>), dim3(a.H, a.B, 1), dim3(1024), 0, s, a); break;
      case 14: hipLaunchKernelGGL((decode_attn_pipe_kernel<8, 1>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
      case 15: hipLaunchKernelGGL((decode_attn_pipe_kernel<8, 2>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
      case 16: hipLaunchKernelGGL((decode_attn_pipe_kernel<8, 4>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
      case 17: hipLaunchKernelGGL((decode_attn_pipe_kernel<4, 2>), dim3(a.H, a.B, 1), dim3(256), 0, s, a); break;
      case 18: hipLaunchKernelGGL((decode_attn_pipe_kernel<4, 4>), dim3(a.H, a.B, 1), dim3(256), 0, s, a); break;
      case 1: hipLaunchKernelGGL((decode_attn_kernel<16, 4>), dim3(a.H, a.B, 1), dim3(1024), 0, s, a); break;
      case 2: hipLaunchKernelGGL((decode_attn_kernel<16, 2>), dim3(a.H, a.B, 1), dim3(1024), 0, s, a); break;
      case 3: hipLaunchKernelGGL((decode_attn_kernel<16, 1>), dim3(a.H, a.B, 1), dim3(1024), 0, s, a); break;
      case 4: hipLaunchKernelGGL((decode_attn_kernel<8, 4>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
      case 5: hipLaunchKernelGGL((decode_attn_kernel<8, 2>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
      case 6: hipLaunchKernelGGL((decode_attn_kernel<8, 1>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
      case 7: hipLaunchKernelGGL((decode_attn_kernel<4, 4>), dim3(a.H, a.B, 1), dim3(256), 0, s, a); break;
      case 8: hipLaunchKernelGGL((decode_attn_kernel<4, 2>), dim3(a.H, a.B, 1), dim3(256), 0, s, a); break;
      case 9: hipLaunchKernelGGL((decode_attn_kernel<4, 8>), dim3(a.H, a.B, 1), dim3(256), 0, s, a); break;
      case 10: hipLaunchKernelGGL((decode_attn_kernel<2, 4>), dim3(a.H, a.B, 1), dim3(128), 0, s, a); break;
      case 11: hipLaunchKernelGGL((decode_attn_kernel<2, 8>), dim3(a.H, a.B, 1), dim3(128), 0, s, a); break;
#endif
      // r3: 8 waves, 4-key steps, the next step in flight while this one is computed (decode_attn_pipe_kernel). Per-wave stamps
      // (tools/decode_bench TS_EXP): the waves that share a SIMD finish in dispatch order -- 16 waves: 16.3 / 17.3 / 17.5 / 18.8 us after
      // entry, 8 waves: 14.7 / 15.3 -- and the cross-wave merge is half as long; the key loop itself streams at 6.4-6.9 TB/s either way.
      default: hipLaunchKernelGGL((decode_attn_pipe_kernel<8, 1>), dim3(a.H, a.B, 1), dim3(512), 0, s, a); break;
    }
  } else {
    // few (row, head) pairs: split every key range so that about one workgroup per CU streams the cache
    int nsplit = 256 / (a.B * a.H);
    nsplit = nsplit < 2 ? 2 : (nsplit > DECODE_NSPLIT ? DECODE_NSPLIT : nsplit);
    switch (variant) {
#ifdef GEMV_ALL_VARIANTS
      case 1: hipLaunchKernelGGL((decode_attn_kernel<4, 2>), dim3(a.H, a.B, nsplit), dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL((decode_attn_kernel<8, 2>), dim3(a.H, a.B, nsplit), dim3(512), 0, s, a); break;
      case 3: hipLaunchKernelGGL((decode_attn_kernel<8, 1>), dim3(a.H, a.B, nsplit), dim3(512), 0, s, a); break;
      case 4: hipLaunchKernelGGL((decode_attn_kernel<16, 1>), dim3(a.H, a.B, nsplit), dim3(1024), 0, s, a); break;
      case 5: hipLaunchKernelGGL((decode_attn_kernel<16, 2>), dim3(a.H, a.B, nsplit), dim3(1024), 0, s, a); break;
      case 6: hipLaunchKernelGGL((decode_attn_kernel<4, 4>), dim3(a.H, a.B, nsplit), dim3(256), 0, s, a); break;
#endif
      case 100: hipLaunchKernelGGL((decode_attn_kernel<16, 1>), dim3(a.H, a.B, nsplit), dim3(1024), 0, s, a); break;   // r2 kernel
      default: hipLaunchKernelGGL((decode_attn_pipe_kernel<16, 1>), dim3(a.H, a.B, nsplit), dim3(1024), 0, s, a); break;
    }
    hipLaunchKernelGGL(decode_attn_combine_kernel, dim3(a.H, a.B), dim3(128), 0, s, a, nsplit);
  }
  return hipGetLastError();
}

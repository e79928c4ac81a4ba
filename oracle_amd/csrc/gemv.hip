// Decode-time skinny GEMM: Y[b][n] = epilogue( sum_k X'[b][k] * W[n][k] ), 1 <= M=B <= 32 rows (16 per MFMA B tile).
// This is the HBM-bound heart of KV-cached greedy decode (13.2 GB of bf16 weights streamed per step,
// shared by all rows of the batch). One workgroup = WAVES waves owns NT n-tiles of 16 weight rows and splits K
// WAVES ways; every wave streams its 1-KiB packed blocks straight from HBM into VGPRs with non-temporal 16-byte
// loads (no LDS on the weight path; a block is read exactly once by exactly one wave), two batches of U k-tiles in
// flight (register ping-pong), and feeds them to v_mfma_f32_16x16x32_bf16 as the A operand with the <=16 activation
// rows as the B operand (read from L2; lanes of unused batch columns load nothing).
// KMODE 0: wave w owns a contiguous run of k-tiles; KMODE 1: k-tiles are dealt round-robin (kt = w + WAVES*i), so at
// any moment a workgroup reads one contiguous WAVES-KiB window of its weight rows.
// Partial sums meet in LDS. Fused epilogues (each removes a launch from the 32-layer decode step):
//   EPI_ROPE_KV     q/k/v projection epilogue applies rotary embedding and writes K,V into the cache
//   EPI_SWIGLU      SiLU(gate) * up ;  EPI_F32 fp32 logits
//   EPI_RESIDUAL    h += y, and (ssq_out) per-n-tile partial sums of squares of the new h, so the RMSNorm that
//                   follows needs no reduction pass (norm.hip rmsnorm_part_kernel sums them in a fixed order)
// RMSNorm itself is NOT fused into the GEMV prologue: every workgroup would re-normalise the whole activation
// (measured +11 us of VALU on the 180 MB gate/up launch vs +3.5 us for the separate small launch).
// Every epilogue reproduces the reference's bf16 rounding points.
#include <hip/hip_ext.h>
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Rows: the MFMA B operand holds 16 batch rows; with RH = 2 "row halves" a workgroup multiplies every weight fragment it has
// loaded by TWO B fragments (rows 0..15 and 16..31), so up to 32 rows share one pass over the weights. A row's arithmetic
// (k order, MFMA sequence, reduction order) is the same for RH = 1 and RH = 2 and for any number of live rows.
// Fragment-ordered activations of more than 16 rows are two 16-row images one after the other (common.h xf_off_rows).

// ---- losslessly compressed weights (CW kernels). bf16 weights of one 16 x 32 fragment block span few binades: the high byte
// of an element (sign, exponent bits 7..1) is stored as 4 bits -- sign and a 3-bit offset d from the block's base eh -- and the
// low byte (exponent bit 0, mantissa) as it is, so a block is 768 B instead of 1 KiB and the decode GEMVs -- HBM-bound on the
// weight stream -- move 25 % fewer bytes. The decode is exact (bit-for-bit the original bf16): a block covers the 16 binades
// [2 eh, 2 eh + 15]; elements outside (zeros / tiny values under a large block maximum: ~8e-5 of Gaussian weights, ~4 % of the
// blocks) are "escaped": the block is flagged, gives up its lowest exponent code as the marker, and an element with that code
// carries in its sign and mantissa fields an 8-bit index into the unit's patch table of true values (256 entries per GEMV
// unit, in LDS). (Tried: a second per-block exponent for the escaped elements, so that only blocks with mixed out-of-range
// exponents need the table -- 10x fewer lookups, but the second side byte and the longer decode made the COMMON path slower:
// gate/up 24.7 -> 25.9 us on Gaussian weights. Removed.)
// Lane layout (12 B): L0, L1 = low bytes of elements 0-3, 4-7; H = the eight 4-bit fields, element k < 4 in nibble 2k,
// element 4 + k in nibble 2k + 1. ~17 integer vector ops per lane and block:
//   four high bytes at once: (x & 0x07070707) + eh * 0x01010101 + ((x & 0x08080808) << 4), then v_perm interleaves.
typedef __attribute__((ext_vector_type(3))) uint32_t u32x3;
struct CwReg { u32x3 c; };
// the block's emin byte through the scalar (constant) path: the address is wave-uniform, so this is an s_load into an SGPR --
// no vector-memory instruction (a per-lane byte load per block, or a per-unit gather, cost 2-4 us per launch) and the base
// replication / escape test below become scalar work. `word` = the aligned dword holding the byte.
typedef __attribute__((address_space(4))) const uint32_t cw_cu32;
__device__ __forceinline__ uint32_t cw_base_word(const uint8_t* emin, size_t blk) {
  return ((cw_cu32*)(uintptr_t)emin)[blk >> 2];
}
__device__ __forceinline__ uint32_t cw_base_byte(uint32_t word, size_t blk) { return (word >> (8 * (blk & 3))) & 0xffu; }
__device__ __forceinline__ uint32_t cw_high4(uint32_t x, uint32_t eh4) {   // four 4-bit fields (one per byte) -> four high bytes
  return ((x & 0x07070707u) + eh4) + ((x & 0x08080808u) << 4);
}
// e: the block's byte of GemvArgs::emin, wave-uniform (an SGPR: the base replication and the escape test are scalar work)
__device__ __forceinline__ bf16x8 cw_decode(const CwReg& r, uint32_t e, const uint16_t* sh_patch) {
  uint32_t eh4 = e & 0x7fu;
  eh4 |= eh4 << 8;
  eh4 |= eh4 << 16;
  const uint32_t ha = cw_high4(r.c[2] & 0x0f0f0f0fu, eh4), hb = cw_high4((r.c[2] >> 4) & 0x0f0f0f0fu, eh4);
  u32x4 t = {__builtin_amdgcn_perm(ha, r.c[0], 0x05010400u), __builtin_amdgcn_perm(ha, r.c[0], 0x07030602u),
             __builtin_amdgcn_perm(hb, r.c[1], 0x05010400u), __builtin_amdgcn_perm(hb, r.c[1], 0x07030602u)};
  if (e & 0x80u) {                                                       // rare (~4 % of blocks): this block has escaped elements
    // which of this lane's eight elements carry exponent code 0? y = the 4-bit codes, one per byte; z = 0x80 in every zero byte
    const uint32_t y0 = (r.c[2] & 0x07070707u) | ((r.c[0] >> 4) & 0x08080808u);
    const uint32_t y1 = ((r.c[2] >> 4) & 0x07070707u) | ((r.c[1] >> 4) & 0x08080808u);
    const uint32_t z0 = ((y0 + 0x7f7f7f7fu) & 0x80808080u) ^ 0x80808080u, z1 = ((y1 + 0x7f7f7f7fu) & 0x80808080u) ^ 0x80808080u;
    uint64_t zz = (uint64_t)z0 | ((uint64_t)z1 << 32);
    while (zz) {                                                         // only the lanes that hold one; usually one lane, one element
      const int k = (__ffsll((unsigned long long)zz) - 1) >> 3;          // element 0..7
      zz &= zz - 1;
      const int sh8 = 8 * (k & 3);
      const uint32_t lowb = ((k < 4 ? r.c[0] : r.c[1]) >> sh8) & 0x7fu;  // patch index: the mantissa field ...
      const uint32_t sgn = (r.c[2] >> (sh8 + (k < 4 ? 3 : 7))) & 1u;     // ... and the sign field
      const uint32_t v = sh_patch[lowb | (sgn << 7)];
      const int word = k >> 1;
      const uint32_t keep = (k & 1) ? 0x0000ffffu : 0xffff0000u, ins = (k & 1) ? v << 16 : v;
#pragma unroll
      for (int p = 0; p < 4; ++p) t[p] = p == word ? (t[p] & keep) | ins : t[p];
    }
  }
  return __builtin_bit_cast(bf16x8, t);
}

// Position of block (ntile, kt) inside Wc (GemvArgs::wc_group). Row-major (G = 0) puts the N / 16 workgroups of a launch on regions a
// fixed stride apart that they walk in step -- which HBM channels are busy at a moment then depends on that stride (K = 11008: the
// down projection ran 2.2 us above the same bytes read interleaved, tools/decode_bench REAL_EXP). Interleaved: the blocks of k-tiles
// [g G, (g + 1) G) of ALL n-tiles lie together (the last group may be shorter), n-tile major inside the group, so the launch reads one
// contiguous window per step whatever N and K are.
__host__ __device__ __forceinline__ size_t cw_block_index(int G, int ntiles, int KT, int ntile, int kt) {
  if (G <= 0) return (size_t)ntile * KT + kt;
  if (G == kCwGroup) {   // the only group size the engine uses: shifts and masks (the kernels' hot loops; wave-uniform scalar work)
    const int k0 = kt & ~(kCwGroup - 1);
    const int Gg = KT - k0 < kCwGroup ? KT - k0 : kCwGroup;
    return (size_t)k0 * ntiles + (size_t)ntile * Gg + (kt - k0);
  }
  const int g = kt / G, k0 = g * G;
  const int Gg = KT - k0 < G ? KT - k0 : G;
  return (size_t)k0 * ntiles + (size_t)ntile * Gg + (kt - k0);
}

// Compressor (orc_engine_finalize, once per matrix): one wave per 1-KiB block of the packed bf16 matrix. nt_unit = n-tiles per
// GEMV unit (the patch table is per unit). cnt[unit] (zeroed by the caller) counts escapes; *overflow is set if a unit
// has more than 254 non-zero ones (the caller then keeps the matrix uncompressed); escaped zeros share two constant entries.
__global__ __launch_bounds__(64) void cw_compress_kernel(const bf16_t* __restrict__ Wp, int KT, int nt_unit, uint8_t* __restrict__ Wc,
                                                         uint8_t* __restrict__ emin, uint16_t* __restrict__ patch,
                                                         int* __restrict__ cnt, int* __restrict__ overflow, int group, int ntiles) {
  const size_t blk = blockIdx.x;
  const int lane = threadIdx.x;
  const u32x4 v = *(const u32x4*)(Wp + blk * 512 + lane * 8);
  uint32_t el[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { el[2 * j] = v[j] & 0xffffu; el[2 * j + 1] = v[j] >> 16; }
  int emax = 0, emn = 255;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ex = (el[j] >> 7) & 0xff;
    emax = ex > emax ? ex : emax;
    emn = ex < emn ? ex : emn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(emax, o, 64), b = __shfl_xor(emn, o, 64);
    emax = a > emax ? a : emax;
    emn = b < emn ? b : emn;
  }
  const int eh = emax > 15 ? (emax - 14) >> 1 : 0;                     // smallest eh with 2 eh + 15 >= emax
  const bool flagged = emn < 2 * eh;                                   // some element lies below the block's 16 binades
  const int unit = (int)(blk / KT) / nt_unit;
  uint32_t lo[2] = {0u, 0u}, hn = 0u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ex = (el[j] >> 7) & 0xff;
    uint32_t low = el[j] & 0xffu, nib;
    if (flagged && ex <= 2 * eh) {                                     // escaped: exponent code 0 + an index into the unit's patch table
      int idx;
      if ((el[j] & 0x7fffu) == 0u) {                                   // +0 / -0 (pruned or padded weights): the two constant entries,
        idx = (el[j] & 0x8000u) ? 0xff : 0x7f;                         // any number of them costs no table space
      } else {
        const int n = atomicAdd(&cnt[unit], 1);
        idx = n < 0x7f ? n : n + 1;                                    // 254 free entries: 0..0x7e, 0x80..0xfe
        if (n < 254) patch[(size_t)unit * 256 + idx] = (uint16_t)el[j];
        else *overflow = 1;
      }
      low = (uint32_t)idx & 0x7fu;
      nib = ((uint32_t)idx >> 4) & 0x8u;
    } else {
      nib = ((el[j] >> 12) & 0x8u) | (uint32_t)((ex - 2 * eh) >> 1);
    }
    lo[j >> 2] |= low << (8 * (j & 3));
    hn |= nib << (j < 4 ? 8 * j : 8 * (j - 4) + 4);
  }
  uint32_t* dst = (uint32_t*)(Wc + cw_block_index(group, ntiles, KT, (int)(blk / KT), (int)(blk % KT)) * 768 + lane * 12);
  dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = hn;
  if (lane == 0) {
    emin[blk] = (uint8_t)(eh | (flagged ? 0x80 : 0));
    if (blk % ((size_t)KT * nt_unit) == 0) {                            // the unit's constant entries
      patch[(size_t)unit * 256 + 0x7f] = 0x0000u;
      patch[(size_t)unit * 256 + 0xff] = 0x8000u;
    }
  }
}

template <bool CW> struct WFrag { bf16x8 v; };
template <> struct WFrag<true> { CwReg r; };

// Row scales of the folded RMSNorm (GemvArgs::ssq_in). The n_part per-tile partials of every row are summed in a fixed
// order that does not depend on the batch: lane (g, b) of scale wave w (the first SW = min(WAVES, 4) waves) owns the
// contiguous slice [(g * SW + w) * per, + per) of row b, per = n_part / (4 * SW) <= 16. The loads are the FIRST loads a wave
// issues (oldest in its in-order queue) and nothing reads them until the wave has finished its first unit, so they cost no
// stall; the slice sums meet in LDS (sh_ps[w][half][lane]) and the epilogue wave adds them w-major, g-minor.
template <int RH> struct ScaleRegs { f32x4 v[RH][4]; };
template <int SW, int RH>
__device__ __forceinline__ void scale_load(const GemvArgs& a, ScaleRegs<RH>& r, int w, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[hf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int b = (lane & 15) + 16 * hf;
    if (!a.ssq_in || w >= SW || b >= a.M) continue;
    const int per = a.n_part / (4 * SW);
    const float* p = a.ssq_in + (size_t)b * a.n_part + (g * SW + w) * per;
    if ((per & 3) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * i < per) r.v[hf][i] = *(const f32x4*)(p + 4 * i);
    } else {   // any other slice length (d_model / 256 not a multiple of 4, e.g. 768, 1280, 2560): element by element
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < per) r.v[hf][j >> 2][j & 3] = p[j];
    }
  }
}
template <int SW, int RH>
__device__ __forceinline__ void scale_park(const GemvArgs& a, const ScaleRegs<RH>& r, float (*sh_ps)[RH][64], int w, int lane) {
  if (!a.ssq_in || w >= SW) return;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += r.v[hf][i][0]; s += r.v[hf][i][1]; s += r.v[hf][i][2]; s += r.v[hf][i][3]; }
    sh_ps[w][hf][lane] = s;
  }
}
template <int SW, int RH>
__device__ __forceinline__ float scale_of_row(const GemvArgs& a, const float (*sh_ps)[RH][64], int hf, int b16) {
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < SW; ++w)
#pragma unroll
    for (int g = 0; g < 4; ++g) s += sh_ps[w][hf][g * 16 + b16];
  return rsqrtf(s / (float)a.norm_d + a.norm_eps);
}

// Operands of the q/k/v epilogue that do not depend on this launch's arithmetic (slot, cached length, rotary row): r2 loaded
// them in the epilogue as a chain of three dependent loads (slot -> length -> cos / sin) at the very end of the kernel, with
// the memory system idle (~2 us of the launch: tools/decode_bench TS_EXP). The token gather of a decode step now leaves them per
// ROW (GemvArgs::row_meta / row_cos / row_sin, misc.hip embed_tokens_kernel) and the pipelined kernel requests them at entry: one
// level of loads, nothing waits for them until the epilogue.
template <int RH> struct RopePre { int sl[RH]; int len[RH]; u32x2 cv[RH], sv[RH]; };
template <int RH>
__device__ __forceinline__ void rope_pre_load(const GemvArgs& a, RopePre<RH>& p, int pair, int lane, bool want_tab) {
  const int g = lane >> 4;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    const int b = (lane & 15) + 16 * hf;
    p.sl[hf] = -1; p.len[hf] = 1;
    p.cv[hf] = u32x2{0u, 0u}; p.sv[hf] = u32x2{0u, 0u};
    if (b >= a.M) continue;
    const u32x2 m = *(const u32x2*)(a.row_meta + 4 * b);               // {slot, cached keys incl. this token, left pad, unfinished}
    p.sl[hf] = (int)m[0]; p.len[hf] = (int)m[1];
    if (want_tab) {
      const int d = (pair & 3) * 16 + 4 * g;
      p.cv[hf] = *(const u32x2*)(a.row_cos + b * 64 + d);
      p.sv[hf] = *(const u32x2*)(a.row_sin + b * 64 + d);
    }
  }
}

// ---- epilogues. `red` = this unit's partial D tiles [WAVES][NT][RH][64 lanes][4]; a lane holds column b = lane & 15 (+ 16 per
// row half) and rows n = 16 * tile + 4 * (lane >> 4) + r of a D tile.
// Paired epilogues (one wave): tile 0 and tile 1 of the unit are partners (gate / up, or rotary halves).
// PRE: the rotary / cache operands were requested at kernel entry (`pre`, by reference so that it stays in registers: a nullable
// pointer sent the whole argument block to scratch memory); otherwise the epilogue loads them itself.
template <int NT, int EPI, int WAVES, int RH, bool PRE = false>
__device__ __forceinline__ void epilogue_pair(const GemvArgs& a, const float (*red)[NT][RH][64][4], int pair, int lane,
                                              const float (*sh_ps)[RH][64], const RopePre<RH>& pre = RopePre<RH>()) {
  static_assert(NT == 2 || (NT == 3 && EPI == EPI_ROPE_KV), "paired tiles (+ one V tile for the q/k/v projection)");
  constexpr int SW = WAVES < 4 ? WAVES : 4;
  const int g = lane >> 4;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    const int b = (lane & 15) + 16 * hf;
    f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
      t0 += *(const f32x4*)&red[i][0][hf][lane][0];
      t1 += *(const f32x4*)&red[i][1][hf][lane][0];
    }
    if (b >= a.M) continue;
    if (a.ssq_in) {   // folded RMSNorm: the row's scale, applied before the first rounding
      const float rs = scale_of_row<SW, RH>(a, sh_ps, hf, lane & 15);
      t0 *= rs;
      t1 *= rs;
    }
    if constexpr (EPI == EPI_SWIGLU) {
      const int n = pair * 16 + 4 * g;
      if (n >= a.Nvalid) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gg = rbf(t0[r]);
        const float s = rbf(gg * sigmoidf_(gg));
        o[r] = s * rbf(t1[r]);
      }
      u32x2 st = {pack2(o[0], o[1]), pack2(o[2], o[3])};
      *(u32x2*)((bf16_t*)a.C + (a.c_xf ? xf_off_rows(b, n, a.ldc) : (size_t)b * a.ldc + n)) = st;
    } else {
      // rows were packed so that tile 0 holds head dims [d0, d0+16) and tile 1 their rotary partners [d0+64, ...)
      const int hs = pair >> 2, d = (pair & 3) * 16 + 4 * g;
      const int sec = hs / a.H, h = hs % a.H;
      const int sl = PRE ? pre.sl[hf] : a.map.slot[b];
      if (sl < 0) continue;                                            // inactive decode row
      int len, pos = 0;
      if constexpr (PRE) {
        len = pre.len[hf];
      } else {
        len = a.cur_len[sl];
        const int pos_raw = len - 1 - a.kbeg[sl];
        pos = pos_raw < 0 ? 0 : (pos_raw >= a.maxpos ? a.maxpos - 1 : pos_raw);
      }
      const int slot = len - 1;
      float lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { lo[r] = rbf(t0[r]); hi[r] = rbf(t1[r]); }
      if (sec < 2) {
        u32x2 cv, sv;
        if constexpr (PRE) {
          cv = pre.cv[hf]; sv = pre.sv[hf];
        } else {
          cv = *(const u32x2*)(a.cos_tab + (size_t)pos * 64 + d);
          sv = *(const u32x2*)(a.sin_tab + (size_t)pos * 64 + d);
        }
        const float c[4] = {lo_bf(cv[0]), hi_bf(cv[0]), lo_bf(cv[1]), hi_bf(cv[1])};
        const float s[4] = {lo_bf(sv[0]), hi_bf(sv[0]), lo_bf(sv[1]), hi_bf(sv[1])};
        float ol[4], oh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ol[r] = rbf(lo[r] * c[r]) + rbf(-hi[r] * s[r]);
          oh[r] = rbf(hi[r] * c[r]) + rbf(lo[r] * s[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { lo[r] = ol[r]; hi[r] = oh[r]; }
      }
      bf16_t* dst;
      if (sec == 0)
        dst = (bf16_t*)a.C + (size_t)b * a.ldc + h * 128 + d;
      else
        dst = (sec == 1 ? a.Kc : a.Vc) + (size_t)sl * a.c_bs + (size_t)h * a.c_hs + (size_t)slot * 128 + d;
      u32x2 s0 = {pack2(lo[0], lo[1]), pack2(lo[2], lo[3])};
      u32x2 s1 = {pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
      *(u32x2*)dst = s0;
      *(u32x2*)(dst + 64) = s1;
    }
  }
}

// Third tile of a q/k/v unit: 16 dims of the V projection (no rotary) -> V cache. Unit u carries V tile u, i.e. dims
// [16 * (u % 8), +16) of head u / 8.
template <int NT, int WAVES, int RH, bool PRE = false>
__device__ __forceinline__ void epilogue_vtile(const GemvArgs& a, const float (*red)[NT][RH][64][4], int unit, int lane,
                                               const float (*sh_ps)[RH][64], const RopePre<RH>& pre = RopePre<RH>()) {
  constexpr int SW = WAVES < 4 ? WAVES : 4;
  const int g = lane >> 4;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    const int b = (lane & 15) + 16 * hf;
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WAVES; ++i) t += *(const f32x4*)&red[i][2][hf][lane][0];
    if (b >= a.M) continue;
    if (a.ssq_in) t *= scale_of_row<SW, RH>(a, sh_ps, hf, lane & 15);
    const int sl = PRE ? pre.sl[hf] : a.map.slot[b];
    if (sl < 0) continue;
    const int h = unit >> 3, d = (unit & 7) * 16 + 4 * g;
    int slot;
    if constexpr (PRE) slot = pre.len[hf] - 1;
    else slot = a.cur_len[sl] - 1;
    u32x2 st = {pack2(t[0], t[1]), pack2(t[2], t[3])};
    *(u32x2*)(a.Vc + (size_t)sl * a.c_bs + (size_t)h * a.c_hs + (size_t)slot * 128 + d) = st;
  }
}

// Un-paired epilogues: tile `nt` of the unit (global n-tile index `ntile`), one wave.
struct NoRes {};
template <int RH> struct ResPre { u32x2 v[RH]; };
template <int NT, int EPI, int WAVES, int RH, typename RP = NoRes>
__device__ __forceinline__ void epilogue_tile(const GemvArgs& a, const float (*red)[NT][RH][64][4], int nt, int ntile, int lane,
                                              const float (*sh_ps)[RH][64], const RP& rpre = RP()) {
  constexpr int SW = WAVES < 4 ? WAVES : 4;
  const int g = lane >> 4;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    const int b = (lane & 15) + 16 * hf;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WAVES; ++i) v += *(const f32x4*)&red[i][nt][hf][lane][0];
    if (a.ssq_in) v *= scale_of_row<SW, RH>(a, sh_ps, hf, lane & 15);   // folded RMSNorm (rows b >= M: sum 0, finite scale)
    const int n = ntile * 16 + 4 * g;
    const bool live = b < a.M && n < a.Nvalid;
    float y[4];
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && live) {
      const u32x2 bv = *(const u32x2*)(a.bias + n);
      bb[0] = lo_bf(bv[0]); bb[1] = hi_bf(bv[0]); bb[2] = lo_bf(bv[1]); bb[3] = hi_bf(bv[1]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = rbf(v[r] + bb[r]);
    if constexpr (EPI == EPI_RESIDUAL) {
      if (live) {
        u32x2 rv;
        if constexpr (sizeof(RP) > 1) rv = rpre.v[hf];                                   // pipelined kernel: requested at entry
        else rv = *(const u32x2*)(a.R + (size_t)b * a.ldr + n);
        y[0] += lo_bf(rv[0]); y[1] += hi_bf(rv[0]); y[2] += lo_bf(rv[1]); y[3] += hi_bf(rv[1]);
      }
      if (a.ssq_out) {   // sum of squares of the stored (bf16) values of this n-tile, per batch row
        float s = 0.f;
        if (live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float t = rbf(y[r]); s += t * t; }
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0 && b < a.M) a.ssq_out[b * (a.N >> 4) + ntile] = s;
      }
    } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = 0.5f * y[r] * (1.0f + erff(y[r] * 0.70710678118654752440f));
    }
    if (!live) continue;
    if constexpr (EPI == EPI_F32) {
      *(f32x4*)((float*)a.C + (size_t)b * a.ldc + n) = f32x4{y[0], y[1], y[2], y[3]};
    } else {
      u32x2 st = {pack2(y[0], y[1]), pack2(y[2], y[3])};
      *(u32x2*)((bf16_t*)a.C + (size_t)b * a.ldc + n) = st;
      if constexpr (EPI == EPI_RESIDUAL) {
        if (a.C_xf) *(u32x2*)(a.C_xf + xf_off_rows(b, n, a.ldc)) = st;
      }
    }
  }
}

// this lane's activation fragment pointers (one per row half) and the k-tile stride
template <int RH>
__device__ __forceinline__ void x_pointers(const GemvArgs& a, int lane, const bf16_t* (&xp)[RH], bool (&row_ok)[RH], int& xstep) {
  const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    const int row = r16 + 16 * hf;
    row_ok[hf] = row < a.M;
    const int rr = row_ok[hf] ? row : 0;
    xp[hf] = a.x_xf ? a.X + (size_t)(rr >> 4) * 16 * a.K + (g * 16 + (rr & 15)) * 8 : a.X + (size_t)rr * a.ldx + g * 8;
  }
  xstep = a.x_xf ? 512 : 32;   // elements between this lane's fragments of consecutive k-tiles
}

template <int NT, int EPI, int WAVES, int U, int KMODE, int RH, bool CW>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(GemvArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WAVES][NT][RH][64][4];
  constexpr int SW = WAVES < 4 ? WAVES : 4;
  __shared__ float sh_ps[SW][RH][64];
  __shared__ uint16_t sh_patch[CW ? 256 : 1];                        // CW: this unit's escaped values
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KT = a.K >> 5;
  const int ntile0 = blockIdx.x * NT;
  ScaleRegs<RH> sc;
  scale_load<SW, RH>(a, sc, w, lane);
  // this wave's k-tiles: kt(i) = k0 + i * kstep, i < nk
  int k0, kstep, nk;
  if constexpr (KMODE == 0) {
    const int per = (KT + WAVES - 1) / WAVES;
    k0 = w * per; kstep = 1;
    nk = KT - k0; nk = nk < 0 ? 0 : (nk > per ? per : nk);
  } else {
    k0 = w; kstep = WAVES;
    nk = (KT - w + WAVES - 1) / WAVES; nk = nk < 0 ? 0 : nk;
  }
  const bf16_t* xp[RH];
  bool row_ok[RH];
  int xstep;
  x_pointers<RH>(a, lane, xp, row_ok, xstep);
  const bf16_t* wp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = a.Wp + (size_t)(ntile0 + nt) * KT * 512 + lane * 8;
  const size_t blk0 = (size_t)ntile0 * KT;                           // CW: first block of this unit

  WFrag<CW> wA[U][NT], wB[U][NT];
  u32x4 xA[U][RH], xB[U][RH];
  uint32_t eA[U][NT], eB[U][NT];                                     // CW: the blocks' emin bytes (SGPRs)
  // defined contents for the slots a short k-run never loads (their MFMAs are skipped by the same uniform branch;
  // leaving them undefined lets the compiler poison the accumulator)
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      eA[u][nt] = eB[u][nt] = 0u;
      if constexpr (CW) { wA[u][nt].r.c = u32x3{0u, 0u, 0u}; wB[u][nt].r = wA[u][nt].r; }
      else wA[u][nt].v = wB[u][nt].v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  auto wload = [&](WFrag<CW> (&dst)[U][NT], u32x4 (&xd)[U][RH], uint32_t (&ed)[U][NT], int i) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u < nk) {   // wave-uniform: a real branch, the loads stay un-waited until their MFMA
        const int kt = k0 + (i + u) * kstep;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (CW) {
            const size_t blk = blk0 + (size_t)nt * KT + kt;
            dst[u][nt].r.c = __builtin_nontemporal_load((const u32x3*)(a.Wc + cw_block_index(a.wc_group, a.N >> 4, KT, ntile0 + nt, kt) * 768 + lane * 12));
            ed[u][nt] = cw_base_byte(cw_base_word(a.emin, blk), blk);
          } else {
            dst[u][nt].v = __builtin_nontemporal_load((const bf16x8*)(wp[nt] + (size_t)kt * 512));
          }
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int hf = 0; hf < RH; ++hf) {
        xd[u][hf] = u32x4{0u, 0u, 0u, 0u};
        if (i + u < nk && row_ok[hf]) xd[u][hf] = *(const u32x4*)(xp[hf] + (size_t)(k0 + (i + u) * kstep) * xstep);
      }
  };
  wload(wA, xA, eA, 0);
  wload(wB, xB, eB, U);
  if constexpr (CW) {   // the unit's patch table, behind the first weight loads
    if (tid < 256) sh_patch[tid] = a.patch[(size_t)blockIdx.x * 256 + tid];
    __syncthreads();
  }

  f32x4 acc[NT][RH];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) acc[nt][hf] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](WFrag<CW> (&src)[U][NT], u32x4 (&xs)[U][RH], uint32_t (&es)[U][NT], int i) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u < nk) {
        bf16x8 wf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (CW) wf[nt] = cw_decode(src[u][nt].r, es[u][nt], sh_patch);
          else wf[nt] = src[u][nt].v;
        }
#pragma unroll
        for (int hf = 0; hf < RH; ++hf) {
          const bf16x8 xf = __builtin_bit_cast(bf16x8, xs[u][hf]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf, acc[nt][hf], 0, 0, 0);
        }
      }
  };

  for (int i = 0; i < nk; i += 2 * U) {
    compute(wA, xA, eA, i);
    wload(wA, xA, eA, i + 2 * U);
    compute(wB, xB, eB, i + U);
    wload(wB, xB, eB, i + 3 * U);
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) *(f32x4*)&red[w][nt][hf][lane][0] = acc[nt][hf];
  scale_park<SW, RH>(a, sc, sh_ps, w, lane);
  __syncthreads();

  if constexpr (EPI == EPI_SWIGLU || EPI == EPI_ROPE_KV) {
    if (w == 0) epilogue_pair<NT, EPI, WAVES, RH>(a, red, blockIdx.x, lane, sh_ps);
    if constexpr (NT == 3) {
      if (w == (WAVES > 1 ? 1 : 0)) epilogue_vtile<NT, WAVES, RH>(a, red, blockIdx.x, lane, sh_ps);
    }
  } else {
    if (w < NT) epilogue_tile<NT, EPI, WAVES, RH>(a, red, w, ntile0 + w, lane, sh_ps);
  }
}

// ---- r3: the same unit / K split / summation order as gemv_kernel<.., KMODE 1, ..> (so the same output bits), but the
// k-loop is a real software pipeline. In gemv_kernel every load sits behind a wave-uniform "is this k-tile mine?" branch and
// the activation loads behind a per-lane row test; hipcc then cannot count its loads and drains the queue (`s_waitcnt
// vmcnt(0)`) at the top of every iteration -- both register buffers are waited for together, nothing is in flight while a
// wave decodes and multiplies -- and the block's side byte (`s_load` + shift) was waited for right where it was requested,
// in front of the activation load (.s of r2). Here:
//   * the main loop runs over FULL steps only (every load unconditional; the ragged end is peeled into one conditional
//     round), so the waits are counted: `vmcnt((NT + RH) * U * (NB - 1))` in front of a buffer's MFMAs;
//   * activation fragments come through a buffer descriptor: lanes of rows >= M carry an out-of-range offset and get zeros
//     from the range check without a branch or a memory request;
//   * the side-byte word stays a raw SGPR until the block is decoded;
//   * NB register buffers of U k-tiles (template), the patch table is requested FIRST (oldest in the queue).
// Host side: needs floor((K / 32) / WAVES) >= NB * U (pick_default falls back to gemv_kernel otherwise: tiny models).
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
#ifndef PIPE_W_AUX
#define PIPE_W_AUX 2   // cache policy of the weight stream's buffer loads (gfx950: bit 0 sc0, bit 1 nt, bit 4 sc1): non-temporal. REAL_EXP: without nt
                       // (0) every launch is 1.3-4 us slower (gate/up 23.7 -> 25.7, lm_head 31.7 -> 35.6); sc0 / sc1 on top of nt change nothing
#endif
template <int NT, int EPI, int WAVES, int U, int RH, bool CW, int NB>
__global__ __launch_bounds__(WAVES * 64) void gemv_pipe_kernel(GemvArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WAVES][NT][RH][64][4];
  constexpr int SW = WAVES < 4 ? WAVES : 4;
  __shared__ float sh_ps[SW][RH][64];
  __shared__ uint16_t sh_patch[CW ? 256 : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KT = a.K >> 5;
  const int ntile0 = blockIdx.x * NT;
#ifdef GEMV_ALL_VARIANTS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  if (a.dbg & 4) ts0 = __builtin_amdgcn_s_memrealtime();
#endif
  ScaleRegs<RH> sc;
  scale_load<SW, RH>(a, sc, w, lane);                                // first loads of the wave (nothing waits for them until the end)
  uint16_t pv = 0;
  if constexpr (CW) {
    if (tid < 256) pv = a.patch[(size_t)blockIdx.x * 256 + tid];      // ahead of the weight loads: its wait leaves them in flight
  }
  // epilogue operands that exist before the launch: requested now by the waves that will run the epilogue
  RopePre<RH> rope_pre;
  ResPre<RH> res_pre;
#pragma unroll
  for (int hf = 0; hf < RH; ++hf) {
    res_pre.v[hf] = u32x2{0u, 0u};
    rope_pre.sl[hf] = -1; rope_pre.len[hf] = 1; rope_pre.cv[hf] = u32x2{0u, 0u}; rope_pre.sv[hf] = u32x2{0u, 0u};
  }
  if constexpr (EPI == EPI_ROPE_KV) {
    if (a.row_meta && w < 2) rope_pre_load<RH>(a, rope_pre, blockIdx.x, lane, w == 0);
  } else if constexpr (EPI == EPI_RESIDUAL) {
    if (w < NT) {
      const int n = (ntile0 + w) * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int hf = 0; hf < RH; ++hf) {
        const int b = (lane & 15) + 16 * hf;
        if (b < a.M && n < a.Nvalid) res_pre.v[hf] = *(const u32x2*)(a.R + (size_t)b * a.ldr + n);
      }
    }
  }
  const int nk = (KT - w + WAVES - 1) / WAVES;                       // k-tiles kt = w + WAVES * i, i < nk (host: nk >= NB * U)
  const int nfull = nk / U;                                          // steps whose U k-tiles all exist
  // activation fragments: buffer loads, lanes of missing rows out of range (-> zeros, no request)
  const uint32_t x_bytes = a.x_xf ? (uint32_t)(((a.M + 15) >> 4) * 16) * (uint32_t)a.K * 2u : (uint32_t)(a.M - 1) * (uint32_t)a.ldx * 2u + (uint32_t)a.K * 2u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (a.dbg & 1) ? 0 : (int)x_bytes, 0x00020000);
  uint32_t xoff[RH];
  {
    const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) {
      const int row = r16 + 16 * hf;
      const uint32_t o = a.x_xf ? ((uint32_t)hf * 16u * (uint32_t)a.K + (uint32_t)(g * 16 + r16) * 8u) * 2u
                                : ((uint32_t)row * (uint32_t)a.ldx + (uint32_t)g * 8u) * 2u;
      xoff[hf] = row < a.M ? o : 0x80000000u;
    }
  }
  const uint32_t xstepB = a.x_xf ? 1024u : 64u;                       // bytes between the fragments of consecutive k-tiles
  const size_t blk0 = (size_t)ntile0 * KT;
  // weights through a descriptor over this unit's blocks (offsets stay inside the unit: 32 bits are plenty)
  const uint32_t w_bytes = (uint32_t)NT * (uint32_t)KT * (CW ? 768u : 1024u);
  // CW with interleaved blocks (wc_group > 0): the descriptor spans the whole matrix (host: < 2 GiB), offsets from cw_block_index
  const bool ilv = CW && a.wc_group > 0;
  const int ntiles_all = a.N >> 4;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      CW ? (void*)(ilv ? a.Wc : a.Wc + blk0 * 768) : (void*)(a.Wp + blk0 * 512), 0,
      (a.dbg & 2) ? 0 : (int)(ilv ? (uint32_t)ntiles_all * (uint32_t)KT * 768u : w_bytes), 0x00020000);
  const int wlane = lane * (CW ? 12 : 16);
  // CW: the side bytes of ALL this wave's blocks, one byte load per n-tile (lane i <- k-tile index i; host: nk <= 64), read
  // back with v_readlane at decode time. (r2 fetched each block's byte by s_load next to its weight load: scalar loads
  // return out of order, so every use drained lgkmcnt -- a scalar-cache round trip per block in the wave's critical path.)
  uint32_t ev[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ev[nt] = 0u;
    if constexpr (CW) {
      if (lane < nk) ev[nt] = a.emin[blk0 + (size_t)nt * KT + w + lane * WAVES];
    }
  }

  WFrag<CW> wb[NB][U][NT];
  u32x4 xb[NB][U][RH];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (CW) wb[b][u][nt].r.c = u32x3{0u, 0u, 0u};
        else wb[b][u][nt].v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int hf = 0; hf < RH; ++hf) xb[b][u][hf] = u32x4{0u, 0u, 0u, 0u};
    }
  auto load1 = [&](WFrag<CW> (&wd)[NT], u32x4 (&xd)[RH], int i) {   // k-tile index i of this wave
    const int kt = w + i * WAVES;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (CW) {
        const int boff = ilv ? (int)cw_block_index(kCwGroup, ntiles_all, KT, ntile0 + nt, kt) : nt * KT + kt;   // host: wc_group is 0 or kCwGroup
        wd[nt].r.c = __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(wrs, wlane, boff * 768, PIPE_W_AUX));
      } else {
        wd[nt].v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, (nt * KT + kt) * 1024, PIPE_W_AUX));
      }
    }
#pragma unroll
    for (int hf = 0; hf < RH; ++hf)
      xd[hf] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)xoff[hf], (int)((uint32_t)kt * xstepB), 0));
  };
  f32x4 acc[NT][RH];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) acc[nt][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mul1 = [&](WFrag<CW> (&ws)[NT], u32x4 (&xs)[RH], int i) {
    bf16x8 wf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (CW) wf[nt] = cw_decode(ws[nt].r, (uint32_t)__builtin_amdgcn_readlane((int)ev[nt], i), sh_patch);
      else wf[nt] = ws[nt].v;
    }
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) {
      const bf16x8 xf = __builtin_bit_cast(bf16x8, xs[hf]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf, acc[nt][hf], 0, 0, 0);
    }
  };

#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int u = 0; u < U; ++u) load1(wb[b][u], xb[b][u], b * U + u);
    __builtin_amdgcn_sched_barrier(0);                                // issue order = the loop's (the counted waits assume it)
  }
  if constexpr (CW) {
    if (tid < 256) sh_patch[tid] = pv;
    __syncthreads();
  }
  int s = 0;
#ifdef GEMV_ALL_VARIANTS
  if (a.dbg & 4) {   // first fragment landed (drains this wave's queue once: timing builds only)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts1 = __builtin_amdgcn_s_memrealtime();
  }
#endif
  for (; s + 2 * NB <= nfull; s += NB) {                              // steps s .. s + 2 NB - 1 are full: no branch around any load
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int u = 0; u < U; ++u) mul1(wb[b][u], xb[b][u], (s + b) * U + u);
#pragma unroll
      for (int u = 0; u < U; ++u) load1(wb[b][u], xb[b][u], (s + NB + b) * U + u);
      __builtin_amdgcn_sched_barrier(0);                              // keep the next buffer's decode behind these loads' issue
    }
  }
  // the ragged end: buffers hold the full steps s .. s + NB - 1; at most NB more steps, some k-tiles missing
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int u = 0; u < U; ++u) mul1(wb[b][u], xb[b][u], (s + b) * U + u);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((s + NB + b) * U + u < nk) load1(wb[b][u], xb[b][u], (s + NB + b) * U + u);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((s + NB + b) * U + u < nk) mul1(wb[b][u], xb[b][u], (s + NB + b) * U + u);

#ifdef GEMV_ALL_VARIANTS
  if (a.dbg & 4) {
    ts2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) a.dbg_ts[(size_t)gridDim.x * 4 + (size_t)blockIdx.x * 16 + w] = ts2 - ts0;   // per wave: entry -> k-loop done
  }
#endif
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) *(f32x4*)&red[w][nt][hf][lane][0] = acc[nt][hf];
  scale_park<SW, RH>(a, sc, sh_ps, w, lane);
  __syncthreads();
  if constexpr (EPI == EPI_ROPE_KV) {
    static_assert(WAVES > 1, "the V tile's epilogue runs on wave 1");
    if (a.row_meta) {
      if (w == 0) epilogue_pair<NT, EPI, WAVES, RH, true>(a, red, blockIdx.x, lane, sh_ps, rope_pre);
      if constexpr (NT == 3) {
        if (w == 1) epilogue_vtile<NT, WAVES, RH, true>(a, red, blockIdx.x, lane, sh_ps, rope_pre);
      }
    } else {
      if (w == 0) epilogue_pair<NT, EPI, WAVES, RH>(a, red, blockIdx.x, lane, sh_ps);
      if constexpr (NT == 3) {
        if (w == 1) epilogue_vtile<NT, WAVES, RH>(a, red, blockIdx.x, lane, sh_ps);
      }
    }
  } else if constexpr (EPI == EPI_SWIGLU) {
    if (w == 0) epilogue_pair<NT, EPI, WAVES, RH>(a, red, blockIdx.x, lane, sh_ps);
  } else if constexpr (EPI == EPI_RESIDUAL) {
    if (w < NT) epilogue_tile<NT, EPI, WAVES, RH, ResPre<RH>>(a, red, w, ntile0 + w, lane, sh_ps, res_pre);
  } else {
    if (w < NT) epilogue_tile<NT, EPI, WAVES, RH>(a, red, w, ntile0 + w, lane, sh_ps);
  }
#ifdef GEMV_ALL_VARIANTS
  if ((a.dbg & 4) && w == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the epilogue's stores have left the wave
    const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* o = a.dbg_ts + (size_t)blockIdx.x * 4;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
    }
  }
#endif
}

// Persistent form: a fixed grid (<= one workgroup per CU), each workgroup walks units u = blockIdx.x, + gridDim.x, ...
// (unit = NT n-tiles) and every wave treats its k-tiles of ALL its units as ONE stream of loads: the register ping-pong
// keeps running across unit boundaries, so the memory pipe of a CU never drains while a unit is reduced and written.
// At a unit boundary each wave parks its partial tiles in LDS (two parities), one barrier, and ONE wave (rotating)
// runs the epilogue while the others are already multiplying the next unit. Bytes in flight per CU are set by
// WAVES * 2 * U * NT KiB and not by how many workgroups happen to share the CU.
template <int NT, int EPI, int WAVES, int U, int RH, bool CW>
__global__ __launch_bounds__(WAVES * 64) void gemv_pers_kernel(GemvArgs a, int units) {
  __shared__ __attribute__((aligned(16))) float red[2][WAVES][NT][RH][64][4];
  constexpr int SW = WAVES < 4 ? WAVES : 4;
  __shared__ float sh_ps[SW][RH][64];
  __shared__ uint16_t sh_patch[2][CW ? 256 : 1];                    // CW: the unit's escaped values (two unit parities)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  ScaleRegs<RH> sc;
  scale_load<SW, RH>(a, sc, w, lane);
  const int KT = a.K >> 5;
  const int nk = (KT - w + WAVES - 1) / WAVES;                       // >= 1 (host: KT >= WAVES)
  const int G = gridDim.x;
  const int nu = (units - (int)blockIdx.x + G - 1) / G;             // >= 1 (host: G <= units)
  const int T = nu * nk;                                            // steps of this wave's stream
  const bf16_t* xp[RH];
  bool row_ok[RH];
  int xstep;
  x_pointers<RH>(a, lane, xp, row_ok, xstep);
  const size_t unit_stride = (size_t)NT * KT * 512;                 // elements between consecutive units
  const bf16_t* wbase = a.Wp + (size_t)blockIdx.x * unit_stride + lane * 8;
  uint16_t pnext = 0;                                               // CW: this thread's patch entry of the NEXT unit, in flight
  auto patch_fetch = [&](int unit) {
    if constexpr (CW) {
      if (tid < 256 && unit < units) pnext = a.patch[(size_t)unit * 256 + tid];
    }
  };

  WFrag<CW> wA[U][NT], wB[U][NT];
  u32x4 xA[U][RH], xB[U][RH];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (CW) { wA[u][nt].r.c = u32x3{0u, 0u, 0u}; wB[u][nt].r = wA[u][nt].r; }
      else wA[u][nt].v = wB[u][nt].v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }

  // load cursor: step ls = (unit lj, k index li)
  int ls = 0, li = 0;
  const bf16_t* lw = wbase;                                         // weights of the load cursor's unit
  size_t lblk = (size_t)blockIdx.x * NT * KT;                       // CW: first block of the load cursor's unit
  uint32_t eA[U][NT], eB[U][NT];                                    // CW: the blocks' emin bytes (SGPRs), loaded with the weights
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) eA[u][nt] = eB[u][nt] = 0u;
  auto wload = [&](WFrag<CW> (&dst)[U][NT], u32x4 (&xd)[U][RH], uint32_t (&ed)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int hf = 0; hf < RH; ++hf) xd[u][hf] = u32x4{0u, 0u, 0u, 0u};
      if (ls < T) {                                                  // wave-uniform
        const int kt = w + li * WAVES;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (CW) {
            const size_t blk = lblk + (size_t)nt * KT + kt;
            dst[u][nt].r.c = __builtin_nontemporal_load((const u32x3*)(a.Wc + cw_block_index(a.wc_group, a.N >> 4, KT, (int)(lblk / KT) + nt, kt) * 768 + lane * 12));
            ed[u][nt] = cw_base_byte(cw_base_word(a.emin, blk), blk);
          } else {
            dst[u][nt].v = __builtin_nontemporal_load((const bf16x8*)(lw + ((size_t)nt * KT + kt) * 512));
          }
        }
#pragma unroll
        for (int hf = 0; hf < RH; ++hf)
          if (row_ok[hf]) xd[u][hf] = *(const u32x4*)(xp[hf] + (size_t)kt * xstep);
        ++ls;
        if (++li == nk) { li = 0; lw += (size_t)G * unit_stride; lblk += (size_t)G * NT * KT; }
      }
    }
  };
  wload(wA, xA, eA);
  wload(wB, xB, eB);
  if constexpr (CW) {   // patch table of the first unit (the weight loads above are already in flight), the second one requested
    patch_fetch(blockIdx.x);
    if (tid < 256) sh_patch[0][tid] = pnext;
    patch_fetch(blockIdx.x + G);
    __syncthreads();
  }

  f32x4 acc[NT][RH];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int hf = 0; hf < RH; ++hf) acc[nt][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cs = 0, ci = 0, cj = 0;                                       // compute cursor
  auto compute = [&](WFrag<CW> (&src)[U][NT], u32x4 (&xs)[U][RH], uint32_t (&es)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (cs < T) {
        bf16x8 wf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (CW) {
            wf[nt] = cw_decode(src[u][nt].r, es[u][nt], sh_patch[cj & 1]);
          } else {
            wf[nt] = src[u][nt].v;
          }
        }
#pragma unroll
        for (int hf = 0; hf < RH; ++hf) {
          const bf16x8 xf = __builtin_bit_cast(bf16x8, xs[u][hf]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf, acc[nt][hf], 0, 0, 0);
        }
        ++cs;
        if (++ci == nk) {                                            // unit cj complete: every wave passes here nu times
          ci = 0;
          float (*rp)[NT][RH][64][4] = red[cj & 1];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int hf = 0; hf < RH; ++hf) {
              *(f32x4*)&rp[w][nt][hf][lane][0] = acc[nt][hf];
              acc[nt][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          if (cj == 0) scale_park<SW, RH>(a, sc, sh_ps, w, lane);  // once, before the first epilogue of the workgroup
          if constexpr (CW) {   // the next unit's patch table (requested one unit ago) into the other parity; request the one after
            if (tid < 256) sh_patch[(cj + 1) & 1][tid] = pnext;
            patch_fetch(blockIdx.x + (cj + 2) * G);
          }
          __syncthreads();
          if (w == cj % WAVES) {
            const int unit = blockIdx.x + cj * G;
            if constexpr (EPI == EPI_SWIGLU || EPI == EPI_ROPE_KV) {
              epilogue_pair<NT, EPI, WAVES, RH>(a, rp, unit, lane, sh_ps);
              if constexpr (NT == 3) epilogue_vtile<NT, WAVES, RH>(a, rp, unit, lane, sh_ps);
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) epilogue_tile<NT, EPI, WAVES, RH>(a, rp, nt, unit * NT + nt, lane, sh_ps);
            }
          }
          ++cj;
        }
      }
  };
  while (cs < T) {
    compute(wA, xA, eA);
    wload(wA, xA, eA);
    compute(wB, xB, eB);
    wload(wB, xB, eB);
  }
}

template <int NT, int EPI, int WAVES, int U, int RH, bool CW>
void go_pers_cw(const GemvArgs& a, int ntiles, int grid, hipStream_t s) {
  if (a.ev_start)
    hipExtLaunchKernelGGL((gemv_pers_kernel<NT, EPI, WAVES, U, RH, CW>), dim3(grid), dim3(WAVES * 64), 0, s, (hipEvent_t)a.ev_start,
                          (hipEvent_t)a.ev_stop, 0, a, ntiles / NT);
  else
    hipLaunchKernelGGL((gemv_pers_kernel<NT, EPI, WAVES, U, RH, CW>), dim3(grid), dim3(WAVES * 64), 0, s, a, ntiles / NT);
}
template <int NT, int EPI, int WAVES, int U, int RH>
void go_pers_rh(const GemvArgs& a, int ntiles, int grid, hipStream_t s) {
  if constexpr (WAVES >= 4) {   // the compressed stream needs 256 threads for the patch table
    if (a.Wc) return go_pers_cw<NT, EPI, WAVES, U, RH, true>(a, ntiles, grid, s);
  }
  go_pers_cw<NT, EPI, WAVES, U, RH, false>(a, ntiles, grid, s);
}
// more than 16 rows: the two-row-half kernels (same weights pass, two B fragments per weight fragment)
template <int NT, int EPI, int WAVES, int U>
void go_pers(const GemvArgs& a, int ntiles, int grid, hipStream_t s) {
  if (a.M > 16) go_pers_rh<NT, EPI, WAVES, U, 2>(a, ntiles, grid, s);
  else go_pers_rh<NT, EPI, WAVES, U, 1>(a, ntiles, grid, s);
}

template <int NT, int EPI, int WAVES, int U, int KMODE, int RH, bool CW>
void go_cw(const GemvArgs& a, int ntiles, hipStream_t s) {
  if (a.ev_start)
    hipExtLaunchKernelGGL((gemv_kernel<NT, EPI, WAVES, U, KMODE, RH, CW>), dim3(ntiles / NT), dim3(WAVES * 64), 0, s, (hipEvent_t)a.ev_start,
                          (hipEvent_t)a.ev_stop, 0, a);
  else
    hipLaunchKernelGGL((gemv_kernel<NT, EPI, WAVES, U, KMODE, RH, CW>), dim3(ntiles / NT), dim3(WAVES * 64), 0, s, a);
}
template <int NT, int EPI, int WAVES, int U, int KMODE, int RH>
void go_rh(const GemvArgs& a, int ntiles, hipStream_t s) {
  if constexpr (WAVES >= 4) {   // the compressed stream needs 256 threads for the patch table
    if (a.Wc) return go_cw<NT, EPI, WAVES, U, KMODE, RH, true>(a, ntiles, s);
  }
  go_cw<NT, EPI, WAVES, U, KMODE, RH, false>(a, ntiles, s);
}
template <int NT, int EPI, int WAVES, int U, int KMODE>
void go(const GemvArgs& a, int ntiles, hipStream_t s) {
  if (a.M > 16) go_rh<NT, EPI, WAVES, U, KMODE, 2>(a, ntiles, s);
  else go_rh<NT, EPI, WAVES, U, KMODE, 1>(a, ntiles, s);
}

// the pipelined one-unit kernel (gemv_pipe_kernel): same K split as go<NT, EPI, WAVES, *, 1>, so the same bits
bool g_gemv_legacy = false;   // tests / tools: force the r2 kernels (launch_gemv_legacy)
template <int WAVES, int U, int NB>
bool pipe_ok(const GemvArgs& a) {
  const int KT = a.K >> 5;
  return !g_gemv_legacy && KT / WAVES >= NB * U && (KT + WAVES - 1) / WAVES <= 64 && (long)a.M * a.ldx * 2 < (1l << 31) &&
         (long)((a.M + 15) / 16) * 16 * a.K * 2 < (1l << 31) && (a.wc_group <= 0 || !a.Wc || (long)(a.N / 16) * KT * 768 < (1l << 31));
}
template <int NT, int EPI, int WAVES, int U, int RH, bool CW, int NB>
void go_pipe_cw(const GemvArgs& a, int ntiles, hipStream_t s) {
  if (a.ev_start)
    hipExtLaunchKernelGGL((gemv_pipe_kernel<NT, EPI, WAVES, U, RH, CW, NB>), dim3(ntiles / NT), dim3(WAVES * 64), 0, s, (hipEvent_t)a.ev_start,
                          (hipEvent_t)a.ev_stop, 0, a);
  else
    hipLaunchKernelGGL((gemv_pipe_kernel<NT, EPI, WAVES, U, RH, CW, NB>), dim3(ntiles / NT), dim3(WAVES * 64), 0, s, a);
}
template <int NT, int EPI, int WAVES, int U, int NB>
void go_pipe(const GemvArgs& a, int ntiles, hipStream_t s) {
  static_assert(WAVES >= 4, "the compressed stream needs 256 threads for the patch table");
  if (a.M > 16) {
    if (a.Wc) go_pipe_cw<NT, EPI, WAVES, U, 2, true, NB>(a, ntiles, s);
    else go_pipe_cw<NT, EPI, WAVES, U, 2, false, NB>(a, ntiles, s);
  } else {
    if (a.Wc) go_pipe_cw<NT, EPI, WAVES, U, 1, true, NB>(a, ntiles, s);
    else go_pipe_cw<NT, EPI, WAVES, U, 1, false, NB>(a, ntiles, s);
  }
}

// Tuned on MI355X with tools/decode_bench (M = 8, LLaMA-7B shapes). Shallow load queues win everywhere (deeper unrolls
// oversubscribe the memory system: -10..-25 %); about 8 MiB in flight over the chip is the sweet spot.
//   o, down (residual epilogue, N = 4096: 256 units)   one unit per workgroup, K split 16 ways
//   qkv + rotary (256 units of 3 tiles: a rotary pair  one unit per workgroup, K split 8 ways, two k-tiles per buffer
//   of q or k + one v tile)                            (384 two-tile units on 4 waves: 18.8 us; 256 balanced units: 18.1 us)
//   gate/up (688), lm_head (1000)                      persistent: ceil(units / 256) units per workgroup, 12 / 8 waves
//                                                      (30.8 -> 29.4 us, 41.4 -> 40.3 us); with the compressed weight stream
//                                                      gate/up runs 16 waves (23.8 us) and lm_head two workgroups per CU (32.3 us)
// Tried and removed: RMSNorm in the prologue of the persistent kernel (row scales from the partials, activation image in
// LDS, rows <= 8; bit-identical to the separate launch). Kernel time +3.2 us (gate/up) / +2.8 us (qkv) against the
// 4.7 us launch it replaces, and end to end 3.32 vs 3.29 ms/token: the small launch overlaps the neighbours' ramp.
template <int NT, int EPI>
hipError_t pick_default(const GemvArgs& a, int ntiles, hipStream_t s) {
  const int units = ntiles / NT, KT = a.K >> 5;
  const int per = (units + 255) / 256;
  const int grid = (units + per - 1) / per;
  // The K split (number of waves) depends on the SHAPE only, never on the row count: the fp32 summation order, and with
  // it every output bit of a row, is the same alone and inside any batch.
  // Compressed weights (a.Wc): the same K split as the bf16 stream (so the same bits), but 768-B blocks need more blocks in
  // flight for the same bytes in flight: two k-tiles per buffer on the short-K one-unit kernels, two workgroups per CU on lm_head.
  if (units <= 256 || KT < 12) {
    // r3: the pipelined kernel wherever the k-run is long enough (pipe_ok), on the r2 kernels' K split (8 waves for the
    // 3-tile q/k/v units, 16 otherwise): identical bits, counted waits.
    if (units <= 256 && EPI == EPI_ROPE_KV && KT >= 16) {
      if (pipe_ok<8, 2, 2>(a)) go_pipe<NT, EPI, 8, 2, 2>(a, ntiles, s);
      else go<NT, EPI, 8, 2, 1>(a, ntiles, s);   // 3-tile units: 18.1 vs 18.4 us
    } else if (units <= 256 && a.Wc && KT >= 64 && KT <= 128 && !pipe_ok<16, 1, 2>(a)) {
      go<NT, EPI, 16, 2, 1>(a, ntiles, s);   // o_proj compressed on the r2 kernel: 7.1 vs 7.4 us with two k-tiles per buffer
    } else if (units <= 256) {   // (o_proj compressed, pipelined, on weights that do not fit the Infinity Cache -- REAL_EXP: one k-tile per buffer 7.24 us, two 7.79)
      if (pipe_ok<16, 1, 2>(a)) go_pipe<NT, EPI, 16, 1, 2>(a, ntiles, s);
      else go<NT, EPI, 16, 1, 1>(a, ntiles, s);
    } else go<NT, EPI, 4, 2, 1>(a, ntiles, s);
  } else if constexpr (EPI == EPI_ROPE_KV) {
    go_pers<NT, EPI, 4, 1>(a, ntiles, units, s);
  } else if constexpr (EPI == EPI_SWIGLU) {
    // Compressed stream: about two 8-wave workgroups per CU (460-500 workgroups, one or two units each; the hardware back-fills)
    // instead of one 16-wave workgroup: more waves per CU for the decode's vector work, and a flagged block's LDS round trip
    // holds 8 waves at the unit barrier instead of 16 (tools/decode_bench: 22.7 vs 23.6 us; 12 waves on 230 workgroups: 24.8).
    // The bf16 fallback keeps the same K split (so the same bits).
    // r3: on the compressed stream the pipelined one-unit kernel (688 workgroups of 8 waves, the hardware back-fills) beats the
    // persistent one on real weights: 21.2 vs 23.6 us on Gaussian weights (tools/decode_bench CW_EXP) -- the persistent loop drains
    // its queue every iteration and pays 1.6 us for the 4 % of flagged blocks (its side bytes are scalar loads waited for in
    // place), the pipelined kernel 0.2 us. Same 8-wave K split, so the same bits. The bf16 stream keeps the persistent kernel
    // (30.6 vs 34.6 us: 1-KiB blocks want fewer, longer streams).
    const int grid2 = units < 480 ? units : 480;
    if (a.Wc && pipe_ok<8, 2, 2>(a)) go_pipe<NT, EPI, 8, 2, 2>(a, ntiles, s);
    else if (a.Wc) go_pers<NT, EPI, 8, 2>(a, ntiles, grid2, s);
    else go_pers<NT, EPI, 8, 1>(a, ntiles, grid2, s);
  } else {
    const int per2 = (units + 511) / 512;
    if (a.Wc && pipe_ok<8, 2, 2>(a)) go_pipe<NT, EPI, 8, 2, 2>(a, ntiles, s);      // lm_head compressed: 30.3 vs 33.4 us (persistent, grid 500)
    else go_pers<NT, EPI, 8, 1>(a, ntiles, a.Wc ? (units + per2 - 1) / per2 : grid, s);   // lm_head compressed: grid 500: 32.3 vs 39.4 us
  }
  return hipGetLastError();
}

template <int NT, int EPI>
hipError_t pick(const GemvArgs& a, int ntiles, int variant, hipStream_t s) {
  if (ntiles % NT) return hipErrorInvalidValue;
  switch (variant) {
#ifdef GEMV_ALL_VARIANTS   // tools/decode_bench: variant = 1 + wi*3 + ui, W in {1,2,4,8,16}, U in {2,4,8}, KMODE 1
    case 7: go<NT, EPI, 4, 2, 1>(a, ntiles, s); break;
    case 8: go<NT, EPI, 4, 4, 1>(a, ntiles, s); break;
    case 9: go<NT, EPI, 4, 8, 1>(a, ntiles, s); break;
    case 10: go<NT, EPI, 8, 2, 1>(a, ntiles, s); break;
    case 11: go<NT, EPI, 8, 4, 1>(a, ntiles, s); break;
    case 12: go<NT, EPI, 8, 8, 1>(a, ntiles, s); break;
    case 13: go<NT, EPI, 16, 2, 1>(a, ntiles, s); break;
    case 14: go<NT, EPI, 16, 4, 1>(a, ntiles, s); break;
    case 15: go<NT, EPI, 16, 1, 1>(a, ntiles, s); break;
    // pipelined kernel: 40 + 6 * (W == 16) + 3 * (U - 1) + (NB - 2), W in {8, 16}, U in {1, 2}, NB in {2, 3, 4}
#define PIPE_CASE(v, W, U_, NB_) case v: if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_ROPE_KV || EPI == EPI_SWIGLU || EPI == EPI_F32) { if ((NT == 2 && EPI == EPI_RESIDUAL) || !pipe_ok<W, U_, NB_>(a)) return hipErrorInvalidValue; go_pipe<NT, EPI, W, U_, NB_>(a, ntiles, s); break; } else return hipErrorInvalidValue;
    PIPE_CASE(40, 8, 1, 2) PIPE_CASE(41, 8, 1, 3) PIPE_CASE(42, 8, 1, 4) PIPE_CASE(43, 8, 2, 2) PIPE_CASE(44, 8, 2, 3) PIPE_CASE(45, 8, 2, 4)
    PIPE_CASE(46, 16, 1, 2) PIPE_CASE(47, 16, 1, 3) PIPE_CASE(48, 16, 1, 4) PIPE_CASE(49, 16, 2, 2) PIPE_CASE(50, 16, 2, 3) PIPE_CASE(51, 16, 2, 4)
#undef PIPE_CASE
#endif
    default:
      return pick_default<NT, EPI>(a, ntiles, s);
  }
  return hipGetLastError();
}

}  // namespace

// variant 0 = tuned default; others (GEMV_ALL_VARIANTS builds only) select (WAVES, U, KMODE) for tools/decode_bench
hipError_t launch_gemv_ex(const GemvArgs& a, int epi, int variant, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (a.M > 32 || (a.K & 31) || (a.N & 15) || (!a.Wp && !a.Wc)) return hipErrorInvalidValue;
  if (a.Wc && (!a.emin || !a.patch)) return hipErrorInvalidValue;
  if (a.Wc && a.wc_group != 0 && a.wc_group != kCwGroup) return hipErrorInvalidValue;
  const int ntiles = a.N / 16;
  switch (epi) {
    case EPI_NONE: return pick<1, EPI_NONE>(a, ntiles, variant, s);
    case EPI_GELU: return pick<1, EPI_GELU>(a, ntiles, variant, s);
#ifdef GEMV_ALL_VARIANTS   // tools/decode_bench: variants 20 + v = two-tile units for the residual projections (128 workgroups at N = 4096)
    case EPI_RESIDUAL: return variant >= 20 && variant < 40 ? pick<2, EPI_RESIDUAL>(a, ntiles, variant - 20, s) : pick<1, EPI_RESIDUAL>(a, ntiles, variant, s);
#else
    case EPI_RESIDUAL: return pick<1, EPI_RESIDUAL>(a, ntiles, variant, s);
#endif
    case EPI_SWIGLU: return pick<2, EPI_SWIGLU>(a, ntiles, variant, s);
    case EPI_ROPE_KV: return pick<3, EPI_ROPE_KV>(a, ntiles, variant, s);
    case EPI_F32: return (ntiles % 2 == 0) ? pick<2, EPI_F32>(a, ntiles, variant, s) : pick<1, EPI_F32>(a, ntiles, variant, s);
  }
  return hipErrorInvalidValue;
}

#ifdef GEMV_ALL_VARIANTS
namespace {
template <int NT, int EPI>
hipError_t pick_pers(const GemvArgs& a, int ntiles, int waves, int u, int grid, hipStream_t s) {
  if (ntiles % NT || grid > ntiles / NT || (a.K >> 5) < waves) return hipErrorInvalidValue;
  const int key = waves * 10 + u;
  switch (key) {
    case 41: go_pers<NT, EPI, 4, 1>(a, ntiles, grid, s); break;
    case 42: go_pers<NT, EPI, 4, 2>(a, ntiles, grid, s); break;
    case 81: go_pers<NT, EPI, 8, 1>(a, ntiles, grid, s); break;
    case 82: go_pers<NT, EPI, 8, 2>(a, ntiles, grid, s); break;
    case 121: go_pers<NT, EPI, 12, 1>(a, ntiles, grid, s); break;
    case 122: go_pers<NT, EPI, 12, 2>(a, ntiles, grid, s); break;
    case 161: if constexpr (NT < 3) { go_pers<NT, EPI, 16, 1>(a, ntiles, grid, s); break; } else return hipErrorInvalidValue;   // LDS
    case 162: if constexpr (NT < 3) { go_pers<NT, EPI, 16, 2>(a, ntiles, grid, s); break; } else return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace
// tools/decode_bench only: the persistent kernel with explicit (waves, k-tiles per buffer, grid)
hipError_t launch_gemv_pers(const GemvArgs& a, int epi, int waves, int u, int grid, hipStream_t s) {
  const int ntiles = a.N / 16;
  switch (epi) {
    case EPI_RESIDUAL: return pick_pers<1, EPI_RESIDUAL>(a, ntiles, waves, u, grid, s);
    case EPI_SWIGLU: return pick_pers<2, EPI_SWIGLU>(a, ntiles, waves, u, grid, s);
    case EPI_ROPE_KV: return pick_pers<3, EPI_ROPE_KV>(a, ntiles, waves, u, grid, s);
    case EPI_F32: return pick_pers<2, EPI_F32>(a, ntiles, waves, u, grid, s);
  }
  return hipErrorInvalidValue;
}
#endif

void gemv_force_legacy(int on) { g_gemv_legacy = on != 0; }

hipError_t launch_gemv(const GemmArgs& g, int epi, hipStream_t s) {
  GemvArgs a = {};
  a.map = identity_map();
  a.X = g.X; a.ldx = g.ldx; a.Wp = g.Wp; a.bias = g.bias; a.R = g.R; a.ldr = g.ldr; a.C = g.C; a.ldc = g.ldc;
  a.M = g.M; a.N = g.N; a.K = g.K; a.Nvalid = g.Nvalid;
  return launch_gemv_ex(a, epi, 0, s);
}

// Lossless 12-bit compression of a packed weight matrix for the decode GEMVs (format: top of this file). nt_unit must be the
// NT the GEMV of this shape / epilogue runs with (gemv_unit_tiles). cnt: n_units ints (zeroed here); *overflow_dev: int flag.
int gemv_unit_tiles(int epi, int N) {
  if (epi == EPI_SWIGLU) return 2;
  if (epi == EPI_ROPE_KV) return 3;
  if (epi == EPI_F32) return ((N / 16) % 2 == 0) ? 2 : 1;
  return 1;
}
hipError_t launch_compress_weight(const bf16_t* Wp, int N, int K, int nt_unit, uint8_t* Wc, uint8_t* emin, uint16_t* patch, int* cnt,
                                  int* overflow_dev, hipStream_t s, int group) {
  if ((N & 15) || (K & 31) || nt_unit < 1 || (N / 16) % nt_unit || group < 0) return hipErrorInvalidValue;
  const int ntiles = N / 16, KT = K / 32, units = ntiles / nt_unit;
  hipError_t err = hipMemsetAsync(cnt, 0, (size_t)units * 4, s);
  if (err != hipSuccess) return err;
  err = hipMemsetAsync(patch, 0, (size_t)units * 256 * 2, s);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(cw_compress_kernel, dim3((unsigned)((size_t)ntiles * KT)), dim3(64), 0, s, Wp, KT, nt_unit, Wc, emin, patch, cnt, overflow_dev, group, ntiles);
  return hipGetLastError();
}

// Decode-time skinny GEMM: Y[b][n] = epilogue( sum_k X'[b][k] * W[n][k] ), 1 <= M=B <= 16 rows.
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
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int NT, int EPI, int WAVES, int U, int KMODE>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(GemvArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WAVES][NT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int KT = a.K >> 5;
  const int ntile0 = blockIdx.x * NT;
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
  const bool row_ok = r16 < a.M;

  const bf16_t* xp = a.X + (size_t)(row_ok ? r16 : 0) * a.ldx + g * 8;
  const bf16_t* wp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = a.Wp + (size_t)(ntile0 + nt) * KT * 512 + lane * 8;

  bf16x8 wA[U][NT], wB[U][NT];
  u32x4 xA[U], xB[U];
  // defined contents for the slots a short k-run never loads (their MFMAs are skipped by the same uniform branch;
  // leaving them undefined lets the compiler poison the accumulator)
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wA[u][nt] = wB[u][nt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  auto wload = [&](bf16x8 (&dst)[U][NT], u32x4 (&xd)[U], int i) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u < nk) {   // wave-uniform: a real branch, the loads stay un-waited until their MFMA
        const int kt = k0 + (i + u) * kstep;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dst[u][nt] = __builtin_nontemporal_load((const bf16x8*)(wp[nt] + (size_t)kt * 512));
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xd[u] = u32x4{0u, 0u, 0u, 0u};
      if (i + u < nk && row_ok) xd[u] = *(const u32x4*)(xp + (k0 + (i + u) * kstep) * 32);
    }
  };
  wload(wA, xA, 0);
  wload(wB, xB, U);

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](bf16x8 (&src)[U][NT], u32x4 (&xs)[U], int i) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u < nk) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, xs[u]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[u][nt], xf, acc[nt], 0, 0, 0);
      }
  };

  for (int i = 0; i < nk; i += 2 * U) {
    compute(wA, xA, i);
    wload(wA, xA, i + 2 * U);
    compute(wB, xB, i + U);
    wload(wB, xB, i + 3 * U);
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) *(f32x4*)&red[w][nt][lane][0] = acc[nt];
  __syncthreads();

  // lane holds column b = r16 (batch row), rows n = 16*tile + 4g + r of the D tile
  const int b = r16;
  if constexpr (EPI == EPI_SWIGLU || EPI == EPI_ROPE_KV) {
    static_assert((EPI != EPI_SWIGLU && EPI != EPI_ROPE_KV) || NT == 2, "paired tiles");
    if (w != 0) return;
    f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
      t0 += *(const f32x4*)&red[i][0][lane][0];
      t1 += *(const f32x4*)&red[i][NT - 1][lane][0];
    }
    if (b >= a.M) return;
    if constexpr (EPI == EPI_SWIGLU) {
      const int n = blockIdx.x * 16 + 4 * g;
      if (n >= a.Nvalid) return;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gg = rbf(t0[r]);
        const float s = rbf(gg * sigmoidf_(gg));
        o[r] = s * rbf(t1[r]);
      }
      u32x2 st = {pack2(o[0], o[1]), pack2(o[2], o[3])};
      *(u32x2*)((bf16_t*)a.C + (size_t)b * a.ldc + n) = st;
    } else {
      // rows were packed so that tile 0 holds head dims [d0, d0+16) and tile 1 their rotary partners [d0+64, ...)
      const int pair = blockIdx.x;
      const int hs = pair >> 2, d = (pair & 3) * 16 + 4 * g;
      const int sec = hs / a.H, h = hs % a.H;
      const int len = *a.cur_len;
      const int pos_raw = len - 1 - a.kbeg[b];
      const int pos = pos_raw < 0 ? 0 : (pos_raw >= a.maxpos ? a.maxpos - 1 : pos_raw);
      const int slot = len - 1;
      float lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { lo[r] = rbf(t0[r]); hi[r] = rbf(t1[r]); }
      if (sec < 2) {
        const u32x2 cv = *(const u32x2*)(a.cos_tab + (size_t)pos * 64 + d);
        const u32x2 sv = *(const u32x2*)(a.sin_tab + (size_t)pos * 64 + d);
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
        dst = (sec == 1 ? a.Kc : a.Vc) + (size_t)b * a.c_bs + (size_t)h * a.c_hs + (size_t)slot * 128 + d;
      u32x2 s0 = {pack2(lo[0], lo[1]), pack2(lo[2], lo[3])};
      u32x2 s1 = {pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
      *(u32x2*)dst = s0;
      *(u32x2*)(dst + 64) = s1;
    }
  } else {
    if (w >= NT) return;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WAVES; ++i) v += *(const f32x4*)&red[i][w][lane][0];
    const int n = (ntile0 + w) * 16 + 4 * g;
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
        const u32x2 rv = *(const u32x2*)(a.R + (size_t)b * a.ldr + n);
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
        if (g == 0) a.ssq_out[b * (a.N >> 4) + ntile0 + w] = s;
      }
    } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = 0.5f * y[r] * (1.0f + erff(y[r] * 0.70710678118654752440f));
    }
    if (!live) return;
    if constexpr (EPI == EPI_F32) {
      *(f32x4*)((float*)a.C + (size_t)b * a.ldc + n) = f32x4{y[0], y[1], y[2], y[3]};
    } else {
      u32x2 st = {pack2(y[0], y[1]), pack2(y[2], y[3])};
      *(u32x2*)((bf16_t*)a.C + (size_t)b * a.ldc + n) = st;
    }
  }
}

template <int NT, int EPI, int WAVES, int U, int KMODE>
void go(const GemvArgs& a, int ntiles, hipStream_t s) {
  hipLaunchKernelGGL((gemv_kernel<NT, EPI, WAVES, U, KMODE>), dim3(ntiles / NT), dim3(WAVES * 64), 0, s, a);
}

template <int NT, int EPI>
hipError_t pick(const GemvArgs& a, int ntiles, int variant, hipStream_t s) {
  if (ntiles % NT) return hipErrorInvalidValue;
  switch (variant) {
#ifdef GEMV_ALL_VARIANTS   // tools/decode_bench: variant = 1 + wi*3 + ui, W in {1,2,4,8,16}, U in {2,4,8}, KMODE 1
    case 1: go<NT, EPI, 1, 2, 1>(a, ntiles, s); break;
    case 2: go<NT, EPI, 1, 4, 1>(a, ntiles, s); break;
    case 3: go<NT, EPI, 1, 8, 1>(a, ntiles, s); break;
    case 4: go<NT, EPI, 2, 2, 1>(a, ntiles, s); break;
    case 5: go<NT, EPI, 2, 4, 1>(a, ntiles, s); break;
    case 6: go<NT, EPI, 2, 8, 1>(a, ntiles, s); break;
    case 7: go<NT, EPI, 4, 2, 1>(a, ntiles, s); break;
    case 8: go<NT, EPI, 4, 4, 1>(a, ntiles, s); break;
    case 9: go<NT, EPI, 4, 8, 1>(a, ntiles, s); break;
    case 10: go<NT, EPI, 8, 2, 1>(a, ntiles, s); break;
    case 11: go<NT, EPI, 8, 4, 1>(a, ntiles, s); break;
    case 12: go<NT, EPI, 8, 8, 1>(a, ntiles, s); break;
    case 13: go<NT, EPI, 16, 2, 1>(a, ntiles, s); break;
    case 14: go<NT, EPI, 16, 4, 1>(a, ntiles, s); break;
    case 15: go<NT, EPI, 16, 1, 1>(a, ntiles, s); break;
#endif
    default:
      // tuned on MI355X with tools/decode_bench (M = 8): N <= 4096 wants the K split 16 ways, large N fewer waves;
      // shallow load queues win in both cases (deeper unrolls oversubscribe the memory system: -10..-25 %)
      if (ntiles / NT <= 256) go<NT, EPI, 16, 1, 1>(a, ntiles, s);
      else go<NT, EPI, 4, 2, 1>(a, ntiles, s);
      break;
  }
  return hipGetLastError();
}

}  // namespace

// variant 0 = tuned default; others (GEMV_ALL_VARIANTS builds only) select (WAVES, U, KMODE) for tools/decode_bench
hipError_t launch_gemv_ex(const GemvArgs& a, int epi, int variant, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (a.M > 16 || (a.K & 31) || (a.N & 15)) return hipErrorInvalidValue;
  const int ntiles = a.N / 16;
  switch (epi) {
    case EPI_NONE: return pick<1, EPI_NONE>(a, ntiles, variant, s);
    case EPI_GELU: return pick<1, EPI_GELU>(a, ntiles, variant, s);
    case EPI_RESIDUAL: return pick<1, EPI_RESIDUAL>(a, ntiles, variant, s);
    case EPI_SWIGLU: return pick<2, EPI_SWIGLU>(a, ntiles, variant, s);
    case EPI_ROPE_KV: return pick<2, EPI_ROPE_KV>(a, ntiles, variant, s);
    case EPI_F32: return (ntiles % 2 == 0) ? pick<2, EPI_F32>(a, ntiles, variant, s) : pick<1, EPI_F32>(a, ntiles, variant, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemv(const GemmArgs& g, int epi, hipStream_t s) {
  GemvArgs a = {};
  a.X = g.X; a.ldx = g.ldx; a.Wp = g.Wp; a.bias = g.bias; a.R = g.R; a.ldr = g.ldr; a.C = g.C; a.ldc = g.ldc;
  a.M = g.M; a.N = g.N; a.K = g.K; a.Nvalid = g.Nvalid;
  return launch_gemv_ex(a, epi, 0, s);
}

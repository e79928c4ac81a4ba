// Decode-time skinny GEMM: Y[b][n] = epilogue( sum_k X[b][k] * W[n][k] ), 1 <= M=B <= 16 rows.
// This is the HBM-bound heart of KV-cached greedy decode (13.2 GB of bf16 weights streamed per step,
// shared by all rows of the batch). One workgroup = 8 waves owns NT n-tiles of 16 weight rows and splits K
// eight ways; every wave streams its contiguous run of 1-KiB packed blocks straight from HBM into VGPRs
// (no LDS on the weight path; a block is read exactly once by exactly one wave) and feeds them to
// v_mfma_f32_16x16x32_bf16 as the A operand, with the <=16 activation rows as the B operand (read from
// L2). Partial sums meet in LDS; the epilogue reproduces the reference's bf16 rounding points.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int GV_WAVES = 8;
constexpr int GV_UNROLL = 4;

template <int NT, int EPI>
__global__ __launch_bounds__(GV_WAVES * 64) void gemv_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float red[GV_WAVES][NT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int KT = a.K >> 5;
  const int ntile0 = blockIdx.x * NT;
  const int per = (KT + GV_WAVES - 1) / GV_WAVES;
  const int kt_beg = w * per;
  const int kt_end = (kt_beg + per < KT) ? kt_beg + per : KT;

  const int brow = r16 < a.M ? r16 : a.M - 1;
  const bf16_t* xp = a.X + (size_t)brow * a.ldx + g * 8;
  const bf16_t* wp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = a.Wp + (size_t)(ntile0 + nt) * KT * 512 + lane * 8;

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kt = kt_beg;
  for (; kt + GV_UNROLL <= kt_end; kt += GV_UNROLL) {
    bf16x8 wf[GV_UNROLL][NT], xf[GV_UNROLL];
#pragma unroll
    for (int u = 0; u < GV_UNROLL; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wf[u][nt] = __builtin_nontemporal_load((const bf16x8*)(wp[nt] + (size_t)(kt + u) * 512));
      xf[u] = *(const bf16x8*)(xp + (kt + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < GV_UNROLL; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][nt], xf[u], acc[nt], 0, 0, 0);
  }
  for (; kt < kt_end; ++kt) {
    const bf16x8 xf = *(const bf16x8*)(xp + kt * 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8 wf = __builtin_nontemporal_load((const bf16x8*)(wp[nt] + (size_t)kt * 512));
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[nt], 0, 0, 0);
    }
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) *(f32x4*)&red[w][nt][lane][0] = acc[nt];
  __syncthreads();

  // lane holds column b = r16 (batch row), rows n = 16*tile + 4g + r of the D tile
  const int b = r16;
  if constexpr (EPI == EPI_SWIGLU) {
    static_assert(EPI != EPI_SWIGLU || NT == 2, "swiglu pairs gate/up tiles");
    if (w != 0) return;
    f32x4 gt = f32x4{0.f, 0.f, 0.f, 0.f}, up = gt;
#pragma unroll
    for (int i = 0; i < GV_WAVES; ++i) {
      gt += *(const f32x4*)&red[i][0][lane][0];
      up += *(const f32x4*)&red[i][NT - 1][lane][0];
    }
    const int n = blockIdx.x * 16 + 4 * g;
    if (b >= a.M || n >= a.Nvalid) return;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gg = rbf(gt[r]);
      const float s = rbf(gg * sigmoidf_(gg));
      o[r] = s * rbf(up[r]);
    }
    u32x2 st = {pack2(o[0], o[1]), pack2(o[2], o[3])};
    *(u32x2*)((bf16_t*)a.C + (size_t)b * a.ldc + n) = st;
  } else {
    if (w >= NT) return;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < GV_WAVES; ++i) v += *(const f32x4*)&red[i][w][lane][0];
    const int n = (ntile0 + w) * 16 + 4 * g;
    if (b >= a.M || n >= a.Nvalid) return;
    float y[4];
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      const u32x2 bv = *(const u32x2*)(a.bias + n);
      bb[0] = lo_bf(bv[0]); bb[1] = hi_bf(bv[0]); bb[2] = lo_bf(bv[1]); bb[3] = hi_bf(bv[1]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = rbf(v[r] + bb[r]);
    if constexpr (EPI == EPI_RESIDUAL) {
      const u32x2 rv = *(const u32x2*)(a.R + (size_t)b * a.ldr + n);
      y[0] += lo_bf(rv[0]); y[1] += hi_bf(rv[0]); y[2] += lo_bf(rv[1]); y[3] += hi_bf(rv[1]);
    } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = 0.5f * y[r] * (1.0f + erff(y[r] * 0.70710678118654752440f));
    }
    if constexpr (EPI == EPI_F32) {
      *(f32x4*)((float*)a.C + (size_t)b * a.ldc + n) = f32x4{y[0], y[1], y[2], y[3]};
    } else {
      u32x2 st = {pack2(y[0], y[1]), pack2(y[2], y[3])};
      *(u32x2*)((bf16_t*)a.C + (size_t)b * a.ldc + n) = st;
    }
  }
}

}  // namespace

hipError_t launch_gemv(const GemmArgs& a, int epi, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (a.M > 16 || (a.K & 31) || (a.N & 15)) return hipErrorInvalidValue;
  const int ntiles = a.N / 16;
  switch (epi) {
    case EPI_NONE:
      if (ntiles % 2 == 0 && ntiles >= 1024)
        hipLaunchKernelGGL((gemv_kernel<2, EPI_NONE>), dim3(ntiles / 2), dim3(GV_WAVES * 64), 0, s, a);
      else
        hipLaunchKernelGGL((gemv_kernel<1, EPI_NONE>), dim3(ntiles), dim3(GV_WAVES * 64), 0, s, a);
      break;
    case EPI_GELU:
      hipLaunchKernelGGL((gemv_kernel<1, EPI_GELU>), dim3(ntiles), dim3(GV_WAVES * 64), 0, s, a);
      break;
    case EPI_RESIDUAL:
      hipLaunchKernelGGL((gemv_kernel<1, EPI_RESIDUAL>), dim3(ntiles), dim3(GV_WAVES * 64), 0, s, a);
      break;
    case EPI_SWIGLU:
      if (ntiles & 1) return hipErrorInvalidValue;
      hipLaunchKernelGGL((gemv_kernel<2, EPI_SWIGLU>), dim3(ntiles / 2), dim3(GV_WAVES * 64), 0, s, a);
      break;
    case EPI_F32:
      if (ntiles % 2 == 0)
        hipLaunchKernelGGL((gemv_kernel<2, EPI_F32>), dim3(ntiles / 2), dim3(GV_WAVES * 64), 0, s, a);
      else
        hipLaunchKernelGGL((gemv_kernel<1, EPI_F32>), dim3(ntiles), dim3(GV_WAVES * 64), 0, s, a);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

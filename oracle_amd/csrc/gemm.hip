// bf16 MFMA GEMM for the compute-bound stages (ViT, pooler, projector, LLaMA prefill) and the
// weight packer. C[m][n] = epilogue( sum_k X[m][k] * W[n][k] ), fp32 accumulation on
// v_mfma_f32_16x16x32_bf16, W in the packed fragment order of common.h.
//
// Tile: 128(m) x 128(n) x 64(k) per 256-thread workgroup (4 waves as 2x2, each 64x64 = 4x4 MFMA tiles),
// double-buffered LDS (2 x 32 KiB), register-staged global->LDS with the loads of tile t+1 issued before
// the MFMAs of tile t and written after them (one barrier per k-tile). X rows are stored row-major in
// LDS with a 16-byte-chunk XOR swizzle (chunk ^ (row & 7)) so the ds_read_b128 fragment reads are
// conflict-free; W blocks are copied linearly (already in fragment order).
// The MFMA takes W as the A operand and X as the B operand, so a lane ends up with 4 consecutive n for
// one m and the epilogue stores 8 bytes per (tile, lane).
//
// Epilogues reproduce the rounding points of the reference's eager bf16 graph (each torch op rounds its
// output to bf16): Linear -> [QuickGELU | GELU(erf) | + residual | SiLU(gate) * up] -> bf16.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int g = lane >> 4, r16 = lane & 15;

  const int MT = (a.M + 127) >> 7;
  // XCD-aware, bijective remap: the 8 XCDs each get a contiguous range of tiles (shared W panel in L2)
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  // grouped raster inside the XCD's range: the ~64 tiles an XCD runs at once form an 8 (m) x 8 (n) block, so its L2
  // serves 8 X panels + 8 W panels per k-step instead of a whole column of X panels (measured: 8.8 GB of fabric
  // reads per 22016x4096 prefill launch with the column-major order, 30x the algorithmic bytes)
  constexpr int GM = 8;
  const int NTt = a.N >> 7;
  const int grp = t / (GM * NTt), rin = t % (GM * NTt);
  const int gm = (MT - grp * GM) < GM ? (MT - grp * GM) : GM;
  const int tm = grp * GM + rin % gm, tn = rin / gm;
  const int m0 = tm << 7, n0 = tn << 7;
  const int KT = a.K >> 5;
  const int nk = a.K >> 6;

  u32x4 xr[4], wr[4];
  const int srow = tid >> 3, schunk = tid & 7;
  const bf16_t* xbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int grow = m0 + srow + 32 * i;
    grow = grow < a.M ? grow : a.M - 1;
    xbase[i] = a.X + (size_t)grow * a.ldx + schunk * 8;
  }
  const bf16_t* wbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bi = i * 4 + w, ntile = bi >> 1, ktile = bi & 1;
    wbase[i] = a.Wp + ((size_t)((n0 >> 4) + ntile) * KT + ktile) * 512 + lane * 8;
  }

  auto gload = [&](int kt64) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = *(const u32x4*)(xbase[i] + kt64 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = *(const u32x4*)(wbase[i] + (size_t)kt64 * 1024);
  };
  auto swrite = [&](int buf) {
    char* bx = smem + buf * 32768;
    char* bw = bx + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 32 * i;
      *(u32x4*)(bx + row * 128 + ((schunk ^ (row & 7)) << 4)) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(bw + (i * 4 + w) * 1024 + lane * 16) = wr[i];
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gload(0);
  swrite(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* bx = smem + buf * 32768;
    const char* bw = bx + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(bw + (((wn * 4 + nt) * 2 + kk) << 10) + lane * 16);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 64 + mt * 16 + r16;
        const int c = kk * 4 + g;
        xf[mt] = *(const bf16x8*)(bx + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
    if (kt + 1 < nk) swrite(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + r16;
    if (m >= a.M) continue;
    if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int n = (n0 >> 1) + wn * 32 + p * 16 + 4 * g;  // output column (N/2 wide)
        if (n >= a.Nvalid) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gt = rbf(acc[2 * p][mt][r]);
          const float up = rbf(acc[2 * p + 1][mt][r]);
          const float s = rbf(gt * sigmoidf_(gt));
          o[r] = s * up;
        }
        u32x2 st = {pack2(o[0], o[1]), pack2(o[2], o[3])};
        *(u32x2*)((bf16_t*)a.C + (size_t)m * a.ldc + n) = st;
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * g;
        if (n >= a.Nvalid) continue;
        float y[4];
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
          const u32x2 bv = *(const u32x2*)(a.bias + n);
          b[0] = lo_bf(bv[0]); b[1] = hi_bf(bv[0]); b[2] = lo_bf(bv[1]); b[3] = hi_bf(bv[1]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = rbf(acc[nt][mt][r] + b[r]);
        if constexpr (EPI == EPI_QUICKGELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t1 = rbf(1.702f * y[r]);
            const float sg = rbf(sigmoidf_(t1));
            y[r] = y[r] * sg;
          }
        } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = 0.5f * y[r] * (1.0f + erff(y[r] * 0.70710678118654752440f));
        } else if constexpr (EPI == EPI_RESIDUAL) {
          const u32x2 rv = *(const u32x2*)(a.R + (size_t)m * a.ldr + n);
          y[0] += lo_bf(rv[0]); y[1] += hi_bf(rv[0]); y[2] += lo_bf(rv[1]); y[3] += hi_bf(rv[1]);
        }
        if constexpr (EPI == EPI_F32) {
          *(f32x4*)((float*)a.C + (size_t)m * a.ldc + n) = f32x4{y[0], y[1], y[2], y[3]};
        } else {
          u32x2 st = {pack2(y[0], y[1]), pack2(y[2], y[3])};
          *(u32x2*)((bf16_t*)a.C + (size_t)m * a.ldc + n) = st;
        }
      }
    }
  }
}

// W[Nsrc][K] row-major (optionally through rowmap: packed row -> source row, -1 = zero row)
// -> packed blocks [Np/16][Kp/32][64 lanes][8].
__global__ void pack_weight_kernel(const bf16_t* __restrict__ W, int Nsrc, int K, const int* __restrict__ rowmap,
                                   bf16_t* __restrict__ Wp, int Np, int Kp) {
  const size_t total = (size_t)Np * (Kp >> 3);
  const int KT = Kp >> 5;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(o & 63);
    const size_t blk = o >> 6;
    const int kt = (int)(blk % KT);
    const int nt = (int)(blk / KT);
    const int pr = nt * 16 + (lane & 15);
    const int k = kt * 32 + (lane >> 4) * 8;
    int src = rowmap ? rowmap[pr] : pr;
    if (src >= Nsrc) src = -1;
    bf16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (src >= 0 && k + j < K) ? W[(size_t)src * K + k + j] : (bf16_t)0;
    u32x4 st = {(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
    *(u32x4*)(Wp + o * 8) = st;
  }
}

template <int EPI>
hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int MT = (a.M + 127) / 128, NT = a.N / 128;
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(MT * NT), dim3(256), 65536, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if ((a.N & 127) || (a.K & 63) || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_NONE: return launch_gemm_t<EPI_NONE>(a, s);
    case EPI_QUICKGELU: return launch_gemm_t<EPI_QUICKGELU>(a, s);
    case EPI_GELU: return launch_gemm_t<EPI_GELU>(a, s);
    case EPI_RESIDUAL: return launch_gemm_t<EPI_RESIDUAL>(a, s);
    case EPI_SWIGLU: return launch_gemm_t<EPI_SWIGLU>(a, s);
    case EPI_F32: return launch_gemm_t<EPI_F32>(a, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_pack_weight(const bf16_t* W, int Nsrc, int K, const int* rowmap, bf16_t* Wp, int Np, int Kp,
                              hipStream_t s) {
  if ((Np & 15) || (Kp & 31)) return hipErrorInvalidValue;
  const size_t total = (size_t)Np * (Kp >> 3);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, s, W, Nsrc, K, rowmap, Wp, Np, Kp);
  return hipGetLastError();
}

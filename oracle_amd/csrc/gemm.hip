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
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "common.h"
#include "kernels.h"

namespace {

thread_local int g_gemm_force = 0;   // per thread (a test hook on one thread cannot change another engine's launches). tools: 0 auto, 1 = 128^2 register-staged kernel, 2 / 3 = DMA kernel with BM 256 / 128

// fp32 sigmoid on the hardware exp / rcp (about 1 ulp each): every use rounds the result to bf16 right away
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

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

// ------------------------------------------------------------------------------------------------------------
// Large-M GEMM (prefill, ViT, pooler, projector): BM x 256 x 64 tiles, 8 waves (2 along m x 4 along n, each
// (BM/2) x 64), one workgroup per CU. Both operands go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPR staging and no ds_write: the 128^2 register-staged kernel above is bound by its ds_write_b128 + ds_read traffic,
// 1.3 LDS cycles per MFMA cycle), two 64 KiB stages:
//     wait my DMA of stage s -> barrier -> issue the DMA of tile k+1 into stage s^1 -> MFMAs on stage s
// so a tile's loads fly during the whole previous k-step (~2k MFMA cycles per SIMD).
// W blocks are already in fragment order: one DMA instruction = one 1-KiB block = a conflict-free ds_read_b128 image.
// X rows land as 128-B rows whose 16-B chunks are XOR-swizzled; LDS-DMA destinations are lane-linear, so the swizzle
// is applied to the per-lane SOURCE address (each instruction still reads 8 whole 128-B lines).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// One BM x 256 output tile. `t` = index of the tile inside its class (all tiles of one height), `ntiles` = tiles in the
// class, rows of the class start at `m_base` and MT m-tiles of BM rows cover it.
template <int BM, int EPI>
__device__ __forceinline__ void gemm_dma_tile(const GemmArgs& a, char* smem, int bid, int nwg, int m_base, int MT) {
  constexpr int XB = BM * 128;              // bytes of the X part of a stage
  constexpr int STAGE = XB + 256 * 128;     // + W part
  constexpr int MTW = BM / 32;              // 16-row m-tiles per wave
  constexpr int XI = BM / 64;               // X DMA instructions per wave per stage (8 rows each)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 2, wn = w & 3;
  const int g = lane >> 4, r16 = lane & 15;

  const int NTt = a.N >> 8;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  constexpr int GM = 4;                     // 32 tiles run per XCD at once: 4 (m) x 8 (n)
  const int grp = t / (GM * NTt), rin = t % (GM * NTt);
  const int gm = (MT - grp * GM) < GM ? (MT - grp * GM) : GM;
  const int tm = grp * GM + rin % gm, tn = rin / gm;
  const int m0 = m_base + tm * BM, n0 = tn << 8;
  const int KT = a.K >> 5, nk = a.K >> 6;

  // per-lane DMA sources
  const bf16_t* xsrc[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (i * 8 + w) * 8 + (lane >> 3);          // row inside the tile; LDS image is lane-linear: [row][chunk']
    int grow = m0 + row;
    grow = grow < a.M ? grow : a.M - 1;
    xsrc[i] = a.X + (size_t)grow * a.ldx + (((lane & 7) ^ (row & 7)) << 3);
  }
  const bf16_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = i * 8 + w;                                  // block b = (n-tile b>>1, k-tile b&1)
    wsrc[i] = a.Wp + ((size_t)((n0 >> 4) + (b >> 1)) * KT + (b & 1)) * 512 + lane * 8;
  }
  // instruction pair i of a stage's DMA (X rows (8 i + w) * 8 .., W block 8 i + w); XI == 4 for BM = 256
  auto dma_pair = [&](int kt64, int stage, int i) {
    char* sx = smem + stage * STAGE;
    char* sw = sx + XB;
    if (i < XI)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(xsrc[i] + kt64 * 64), (lds_void_t*)(sx + (i * 8 + w) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc[i] + (size_t)kt64 * 1024), (lds_void_t*)(sw + (i * 8 + w) * 1024), 16, 0, 0);
  };
  auto dma = [&](int kt64, int stage) {
    char* sx = smem + stage * STAGE;
    char* sw = sx + XB;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(xsrc[i] + kt64 * 64), (lds_void_t*)(sx + (i * 8 + w) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc[i] + (size_t)kt64 * 1024), (lds_void_t*)(sw + (i * 8 + w) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][MTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my DMA of this stage has landed
    __syncthreads();                                          // everyone's has, and everyone left the other stage
    // BM = 256: the next tile's 8 DMA instructions go out in pairs behind the first four m-tiles' MFMAs instead of as
    // one burst behind the barrier. Every CU bursting 64 KB at the same moment left 49 % of wave time in s_waitcnt
    // (PMC); spread, the same loads cost 8-12 % less kernel time. BM = 128 has half the MFMAs per k-step to hide a
    // later issue behind and is 10 % slower spread, so it keeps the burst.
    constexpr bool kSpread = MTW == 8;
    if (!kSpread && kt + 1 < nk) dma(kt + 1, st ^ 1);
    const char* sx = smem + st * STAGE;
    const char* sw = sx + XB;
    // W fragments of the whole k-step stay in registers; X fragments stream one 16-row m-tile at a time, the next
    // m-tile's two reads issued ahead of the current tile's 8 MFMAs (register double buffer)
    bf16x8 wf[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[kk][nt] = *(const bf16x8*)(sw + ((((wn * 4 + nt) << 1) + kk) << 10) + lane * 16);
    auto xread = [&](int mt, int kk) -> bf16x8 {
      const int row = wm * (BM / 2) + mt * 16 + r16;
      return *(const bf16x8*)(sx + row * 128 + (((kk * 4 + g) ^ (row & 7)) << 4));
    };
    bf16x8 xc[2] = {xread(0, 0), xread(0, 1)};
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);       // the 8 W + 2 X reads above
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      bf16x8 xn[2] = {xc[0], xc[1]};
      if (mt + 1 < MTW) { xn[0] = xread(mt + 1, 0); xn[1] = xread(mt + 1, 1); }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][nt], xc[kk], acc[nt][mt], 0, 0, 0);
      xc[0] = xn[0]; xc[1] = xn[1];
      // issue order per m-tile: the next tile's two reads, then this tile's 8 MFMAs (otherwise the scheduler sinks the
      // reads to just before their use and waits on them with the matrix pipe idle)
      if (mt + 1 < MTW) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      if (kSpread && mt < 4 && kt + 1 < nk) {
        dma_pair(kt + 1, st ^ 1, mt);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
    }
  }

  // ---- epilogue (same rounding points as gemm_bf16_kernel). bf16 outputs are transposed through LDS (each wave its own
  // (BM/2) x 64 tile, rows padded to 144 B) so global stores and the residual loads are whole 128-B lines, 16 B per lane,
  // instead of 8-B fragments of 16 different rows.
  if constexpr (EPI == EPI_F32) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int m = m0 + wm * (BM / 2) + mt * 16 + r16;
      if (m >= a.M) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * g;
        if (n >= a.Nvalid) continue;
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
          const u32x2 bv = *(const u32x2*)(a.bias + n);
          b[0] = lo_bf(bv[0]); b[1] = hi_bf(bv[0]); b[2] = lo_bf(bv[1]); b[3] = hi_bf(bv[1]);
        }
        *(f32x4*)((float*)a.C + (size_t)m * a.ldc + n) =
            f32x4{rbf(acc[nt][mt][0] + b[0]), rbf(acc[nt][mt][1] + b[1]), rbf(acc[nt][mt][2] + b[2]), rbf(acc[nt][mt][3] + b[3])};
      }
    }
  } else {
    constexpr int ROWB = 144;
    constexpr int WCOLS = EPI == EPI_SWIGLU ? 32 : 64;        // output columns of this wave's tile
    __syncthreads();                                          // every wave has left the stage buffers
    char* tile = smem + w * (BM / 2) * ROWB;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int ml = mt * 16 + r16;
      if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gt = rbf(acc[2 * p][mt][r]);
            const float up = rbf(acc[2 * p + 1][mt][r]);
            const float sg = rbf(gt * sigmoidf_(gt));
            o[r] = sg * up;
          }
          *(u32x2*)(tile + ml * ROWB + (p * 16 + 4 * g) * 2) = u32x2{pack2(o[0], o[1]), pack2(o[2], o[3])};
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int n = n0 + wn * 64 + nt * 16 + 4 * g;
          float y[4];
          float b[4] = {0.f, 0.f, 0.f, 0.f};
          if (a.bias && n < a.Nvalid) {
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
          }
          *(u32x2*)(tile + ml * ROWB + (nt * 16 + 4 * g) * 2) = u32x2{pack2(y[0], y[1]), pack2(y[2], y[3])};
        }
      }
    }
    // rows of the tile back out: 16 B per lane, a wave instruction = (64 / (WCOLS / 8)) rows x one whole row segment
    constexpr int CPR = WCOLS / 8;                             // 16-B chunks per row
    const int ncol0 = (EPI == EPI_SWIGLU ? (n0 >> 1) + wn * 32 : n0 + wn * 64) + (lane % CPR) * 8;
#pragma unroll
    for (int it = 0; it < (BM / 2) * CPR / 64; ++it) {
      const int ml = it * (64 / CPR) + lane / CPR;
      const int m = m0 + wm * (BM / 2) + ml;
      if (m >= a.M || ncol0 >= a.Nvalid) continue;
      u32x4 v = *(const u32x4*)(tile + ml * ROWB + (lane % CPR) * 16);
      if constexpr (EPI == EPI_RESIDUAL) {
        const u32x4 rv = *(const u32x4*)(a.R + (size_t)m * a.ldr + ncol0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = pack2(lo_bf(v[j]) + lo_bf(rv[j]), hi_bf(v[j]) + hi_bf(rv[j]));
      }
      *(u32x4*)((bf16_t*)a.C + (size_t)m * a.ldc + ncol0) = v;
    }
  }
}

// Mixed-height launch: workgroups [0, n_big) compute 256-row tiles of rows [0, m_split), the rest 128-row tiles of rows
// [m_split, M). The hardware hands the short tiles to CUs as they come free, so the partial last round of the tall tiles
// is back-filled instead of idling (384 tall tiles on 256 CUs = 2 rounds, the second half empty: o_proj / down at 5912 rows).
template <int EPI>
__global__ __launch_bounds__(512) void gemm_dma_kernel(GemmArgs a, int n_big, int m_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  if (bid < n_big) {
    gemm_dma_tile<256, EPI>(a, smem, bid, n_big, 0, n_big / (a.N >> 8));
  } else {
    const int nwg = gridDim.x - n_big;
    gemm_dma_tile<128, EPI>(a, smem, bid - n_big, nwg, m_split, nwg / (a.N >> 8));
  }
}

// W[Nsrc][K] row-major (optionally through rowmap: packed row -> source row, -1 = zero row)
// -> packed blocks [Np/16][Kp/32][64 lanes][8].
__global__ void pack_weight_kernel(const bf16_t* __restrict__ W, int Nsrc, int K, const int* __restrict__ rowmap,
                                   bf16_t* __restrict__ Wp, int Np, int Kp, const bf16_t* __restrict__ colscale) {
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
    for (int j = 0; j < 8; ++j) {
      v[j] = (src >= 0 && k + j < K) ? W[(size_t)src * K + k + j] : (bf16_t)0;
      if (colscale && k + j < K) v[j] = f2bf(bf2f(v[j]) * bf2f(colscale[k + j]));
    }
    u32x4 st = {(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
    *(u32x4*)(Wp + o * 8) = st;
  }
}

// Number of 256-row m-tiles (the rest of the rows gets 128-row tiles) that minimises the makespan of greedy in-order
// dispatch on 256 CUs, tall tiles first. Costs in integer units: tall tile 100, short tile 58 (tools/gemm_bench).
// Prompt lengths differ from call to call, so this runs for many (M, N): the dispatch is evaluated in closed form over
// groups of equally loaded CUs (a few dozen steps per candidate) and only splits inside the last tall round are tried.
static long greedy_makespan(long n_tall, long n_short) {
  constexpr int kCUs = 256, kTall = 100, kShort = 58;
  std::map<long, long> load;                         // load -> number of CUs carrying it
  const long q = n_tall / kCUs, r = n_tall % kCUs;
  if (r) load[(q + 1) * kTall] = r;
  load[q * kTall] += kCUs - r;
  long span = (q + (r ? 1 : 0)) * kTall;
  while (n_short > 0) {
    auto lo = load.begin();
    const long l = lo->first, c = lo->second, take = n_short < c ? n_short : c;
    if (take == c) load.erase(lo); else lo->second = c - take;
    load[l + kShort] += take;
    if (l + kShort > span) span = l + kShort;
    n_short -= take;
  }
  return span;
}

static int plan_tall_tiles(int M, int N) {
  static std::map<std::pair<int, int>, int> cache;   // shared by every engine of the process: guarded
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find({M, N});
  if (it != cache.end()) return it->second;
  const int NT = N / 256, max_tall = (M + 255) / 256;
  const int window = 256 / NT + 2;                   // m-tiles that make up one round of tall tiles, and a little
  long best = -1;
  int best_a = max_tall;
  for (int a = max_tall; a >= 0 && a >= max_tall - window; --a) {
    const int rows_b = M - a * 256;
    if (a < max_tall && rows_b <= 0) continue;
    const int mtb = rows_b > 0 ? (rows_b + 127) / 128 : 0;
    const long span = greedy_makespan((long)a * NT, (long)mtb * NT);
    if (best < 0 || span < best) { best = span; best_a = a; }
  }
  if (cache.size() > 4096) cache.clear();
  cache[{M, N}] = best_a;
  return best_a;
}

template <int EPI>
hipError_t launch_dma_t(const GemmArgs& a, int tall, hipStream_t s) {   // tall = number of 256-row m-tiles
  static bool attr_set = false;
  constexpr int stages = 2 * (256 * 128 + 256 * 128), epi = 8 * 128 * 144;
  constexpr int lds = stages > epi ? stages : epi;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int NT = a.N / 256;
  const int m_split = tall * 256 < a.M ? tall * 256 : a.M;
  const int rows_b = a.M - m_split;
  const int mtb = rows_b > 0 ? (rows_b + 127) / 128 : 0;
  hipLaunchKernelGGL((gemm_dma_kernel<EPI>), dim3((tall + mtb) * NT), dim3(512), lds, s, a, tall * NT, m_split);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
  if ((a.M > 128 || g_gemm_force >= 2) && (a.N & 255) == 0 && g_gemm_force != 1) {
    // force 2: 256-row tiles only, 3: 128-row tiles only, 0: the mix that plan_tall_tiles picks
    if (g_gemm_force >= 100) return launch_dma_t<EPI>(a, g_gemm_force - 100 < (a.M + 255) / 256 ? g_gemm_force - 100 : (a.M + 255) / 256, s);   // tools/gemm_bench: explicit count
    const int tall = g_gemm_force == 2 ? (a.M + 255) / 256 : (g_gemm_force == 3 ? 0 : plan_tall_tiles(a.M, a.N));
    return launch_dma_t<EPI>(a, tall, s);
  }
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

void gemm_force_kernel(int which) { g_gemm_force = which; }

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
                              hipStream_t s, const bf16_t* colscale) {
  if ((Np & 15) || (Kp & 31)) return hipErrorInvalidValue;
  const size_t total = (size_t)Np * (Kp >> 3);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, s, W, Nsrc, K, rowmap, Wp, Np, Kp, colscale);
  return hipGetLastError();
}

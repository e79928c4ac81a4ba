// Micro-benchmark of the decode-step kernels (GPU box only; not part of the product or the tests).
// Streams DISTINCT weights on every launch (pool >> 256 MiB Infinity Cache) so the numbers are HBM numbers.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/decode_bench tools/decode_bench.hip \
//          oracle_amd/csrc/gemv.hip oracle_amd/csrc/attn.hip oracle_amd/csrc/norm.hip -DGEMV_ALL_VARIANTS
//   run:   tools/decode_bench [M=8] [ctx=800]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../oracle_amd/csrc/common.h"
#include "../oracle_amd/csrc/kernels.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// plain streaming read: the ceiling for "read B bytes once" at a given launch shape
__global__ __launch_bounds__(512) void read_kernel(const u32x4* __restrict__ p, size_t n16, uint32_t* out) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
    const u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(p + i);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // bf16 in about [-0.03, 0.03]: exponent 0x78..0x79
    p[i] = (uint16_t)(((h & 1) << 15) | (0x3C00 + (h >> 8 & 0xFF)));
  }
}

// N(0, 0.02^2) bf16 values (Box-Muller on a hash): the escape statistics of real weights for the compressed-stream timing
__global__ void gauss_kernel(uint16_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    uint32_t g = h * 747796405u + 2891336453u; g ^= g >> 15; g *= 2246822519u; g ^= g >> 13;
    const float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f), u2 = (g >> 8) * (1.0f / 16777216.0f);
    const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * 0.02f;
    p[i] = (uint16_t)(__float_as_uint(z) >> 16);
  }
}

__global__ void clear_flags_kernel(uint8_t* e, size_t nblk) {   // timing experiment: same bases, no block flagged
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += (size_t)gridDim.x * blockDim.x) e[i] &= 0x7f;
}
// weight distributions a trained model shows (DIST_EXP): kind 0 Gaussian sigma 0.02, 1 Student-t (3 degrees of freedom), 2 outlier
// channels (8 input channels x 30, N / 1024 output rows x 8; kind 5: rows x 20), 3 log-normal row scales (sigma 0.5 in ln: 16 rows of a block spread over ~4 binades), 4 Student-t with 2 degrees of freedom
__device__ __forceinline__ float hash_normal(uint32_t a, uint32_t b) {
  uint32_t h = a * 2654435761u + b * 40503u + 0x9e3779b9u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  uint32_t g = h * 747796405u + 2891336453u; g ^= g >> 15; g *= 2246822519u; g ^= g >> 13;
  const float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f), u2 = (g >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}
__global__ void dist_kernel(uint16_t* p, int N, int K, int kind, uint32_t seed) {
  const size_t n = (size_t)N * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    // p is a PACKED matrix (common.h): element i -> block (nt, kt), lane, j -> logical row / column
    const size_t blk = i >> 9; const int lane = (int)((i >> 3) & 63), j = (int)(i & 7), KT = K >> 5;
    const int r = (int)(blk / KT) * 16 + (lane & 15), c = (int)(blk % KT) * 32 + (lane >> 4) * 8 + j;
    float z = hash_normal((uint32_t)i, seed) * 0.02f;
    if (kind == 1 || kind == 4) {
      const int df = kind == 1 ? 3 : 2;
      float chi = 0.f;
      for (int j = 0; j < df; ++j) { const float t = hash_normal((uint32_t)i, seed + 101u * (j + 1)); chi += t * t; }
      z *= rsqrtf(chi / df + 1e-12f);
    } else if (kind == 2) {
      if ((c * 2654435761u >> 20) % 512 == 7 && c % 7 == 0) z *= 30.0f;   // ~8 of 4096 channels
      if ((r * 2246822519u >> 18) % 1024 == 3) z *= 8.0f;                 // ~ N / 1024 rows
    } else if (kind == 5) {                                               // the same with rows x 20: a whole 16-row tile lifted 4-5 binades
      if ((c * 2654435761u >> 20) % 512 == 7 && c % 7 == 0) z *= 30.0f;
      if ((r * 2246822519u >> 18) % 1024 == 3) z *= 20.0f;
    } else if (kind == 3) {
      z *= expf(hash_normal((uint32_t)r, seed ^ 0x5bd1e995u) * 0.5f);
    }
    p[i] = (uint16_t)((__float_as_uint(z) + 0x7fffu + ((__float_as_uint(z) >> 16) & 1u)) >> 16);
  }
}
struct Shape { const char* name; int N, K, epi; };
int gemv_unit_tiles(int epi, int N);

hipError_t launch_gemv_pers(const GemvArgs& a, int epi, int waves, int u, int grid, hipStream_t s);

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8;
  const int ctx = argc > 2 ? atoi(argv[2]) : 800;
  const int d = 4096, ff = 11008, H = 32;
  const size_t pool_bytes = (size_t)3 << 30;
  bf16_t* pool;
  CK(hipMalloc((void**)&pool, pool_bytes));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, pool, pool_bytes / 2, 1u);
  bf16_t *X, *Y, *R, *lnw, *cosb, *sinb;
  float* ssq;
  CK(hipMalloc((void**)&X, (size_t)16 * ff * 2));
  CK(hipMalloc((void**)&Y, (size_t)16 * 32000 * 4));
  CK(hipMalloc((void**)&R, (size_t)16 * ff * 2));
  CK(hipMalloc((void**)&lnw, (size_t)ff * 2));
  CK(hipMalloc((void**)&cosb, (size_t)4096 * 64 * 2));
  CK(hipMalloc((void**)&sinb, (size_t)4096 * 64 * 2));
  CK(hipMalloc((void**)&ssq, (size_t)16 * 4096 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, X, (size_t)16 * ff, 2u);
  hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, R, (size_t)16 * ff, 3u);
  hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, lnw, (size_t)ff, 4u);
  hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, cosb, (size_t)4096 * 64, 5u);
  hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, sinb, (size_t)4096 * 64, 6u);
  CK(hipMemset(ssq, 0, (size_t)16 * 4096 * 4));
  // KV cache for the attention test + the rope epilogue: [B][H][Tmax][128]
  const int Tmax = ctx + 64;
  bf16_t *Kc, *Vc;
  const size_t kv_elems = (size_t)16 * H * Tmax * 128;
  CK(hipMalloc((void**)&Kc, kv_elems * 2 * 8));   // 8 "layers" to rotate through
  CK(hipMalloc((void**)&Vc, kv_elems * 2 * 8));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, Kc, kv_elems * 8, 7u);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, Vc, kv_elems * 8, 8u);
  int* ints;
  CK(hipMalloc((void**)&ints, 64 * 4));
  {
    int h[64] = {0};
    for (int b = 0; b < 16; ++b) h[b] = ctx;   // cur_len per row; kbeg (ints + 16) = 0
    CK(hipMemcpy(ints, h, sizeof(h), hipMemcpyHostToDevice));
  }
  // per-row operands of the q/k/v epilogue as the engine's token gather leaves them (GemvArgs::row_meta / row_cos / row_sin)
  int* row_meta;
  CK(hipMalloc((void**)&row_meta, 32 * 4 * 4));
  {
    int h[128];
    for (int b = 0; b < 32; ++b) { h[4 * b] = b < 16 ? b : -1; h[4 * b + 1] = ctx; h[4 * b + 2] = 0; h[4 * b + 3] = 1; }
    CK(hipMemcpy(row_meta, h, sizeof(h), hipMemcpyHostToDevice));
  }
  const bf16_t* row_cos = cosb + (size_t)(ctx - 1) * 64;   // every row at position ctx - 1: 32 x 64 entries from there on are readable
  const bf16_t* row_sin = sinb + (size_t)(ctx - 1) * 64;
  const bool use_meta = !getenv("NO_META");
  float* part;
  CK(hipMalloc((void**)&part, (size_t)16 * H * DECODE_NSPLIT * 130 * 4));
  CK(hipDeviceSynchronize());

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t s = 0;
  const int iters = 40;

  auto time_it = [&](auto&& launch) -> float {
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch(i + 3);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
  };

  printf("M=%d ctx=%d  (us per launch incl. ~1.5us boundary; GB/s = algorithmic weight bytes / time)\n", M, ctx);
  // ---- read ceiling
  for (size_t mb : {33, 90, 100, 180}) {
    if (getenv("DBG_EXP") || getenv("TS_EXP") || getenv("DIST_EXP")) break;
    const size_t bytes = mb << 20;
    for (int grid : {256, 512, 1024, 2048}) {
      float us = time_it([&](int i) {
        const size_t off = ((size_t)i * bytes) % (pool_bytes - bytes);
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(512), 0, s, (const u32x4*)((char*)pool + (off & ~(size_t)255)), bytes / 16, (uint32_t*)Y);
      });
      printf("read %4zuMB grid %4d : %7.2f us  %7.1f GB/s\n", mb, grid, us, bytes / us * 1e-3);
    }
  }

  const Shape shapes[] = {
      {"qkv  ropekv    ", 3 * d, d, EPI_ROPE_KV},
      {"o    residual  ", d, d, EPI_RESIDUAL},
      {"gu   swiglu    ", 2 * ff, d, EPI_SWIGLU},
      {"down residual  ", d, ff, EPI_RESIDUAL},
      {"head f32       ", 32000, d, EPI_F32},
  };
  for (const Shape& sh : shapes) {
    if (getenv("DBG_EXP") || getenv("TS_EXP") || getenv("DIST_EXP")) break;
    const size_t wbytes = (size_t)sh.N * sh.K * 2;
    for (int variant = 0; variant <= 51; ++variant) {
      if (variant > 15 && variant < 40) continue;                    // 40..51: pipelined kernel (W, U, NB)
      if (getenv("QUICK") && variant != 0 && variant < 40) continue;
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.bias = nullptr; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : (sh.epi == EPI_ROPE_KV ? d : sh.N);
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.ssq_out = ssq; a.x_xf = 1; a.c_xf = 1;
      if (use_meta) { a.row_meta = row_meta; a.row_cos = row_cos; a.row_sin = row_sin; }
      hipError_t err = hipSuccess;
      float us = time_it([&](int i) {
        const size_t off = ((size_t)i * wbytes) % (pool_bytes - wbytes);
        a.Wp = (const bf16_t*)((char*)pool + (off & ~(size_t)1023));
        hipError_t e = launch_gemv_ex(a, sh.epi, variant, s);
        if (e != hipSuccess) err = e;
      });
      if (err != hipSuccess) { printf("%s v%d : %s\n", sh.name, variant, hipGetErrorString(err)); (void)hipGetLastError(); continue; }
      const int vi = variant - 1;
      if (variant == 0) printf("%s default    : %7.2f us  %7.1f GB/s\n", sh.name, us, wbytes / us * 1e-3);
      else if (variant >= 40) printf("%s pipe W%-2d U%d NB%d : %7.2f us  %7.1f GB/s\n", sh.name, variant >= 46 ? 16 : 8, ((variant - 40) % 6) / 3 + 1, (variant - 40) % 3 + 2, us, wbytes / us * 1e-3);
      else printf("%s W%-2d U%d    : %7.2f us  %7.1f GB/s\n", sh.name, 1 << (vi / 3), variant == 15 ? 1 : 2 << (vi % 3), us, wbytes / us * 1e-3);
    }
  }
  // ---- the compressed stream on non-Gaussian weights: escapes, flagged blocks, overflow and time per launch (one matrix per run, so
  // partly cache-resident: compare the rows with each other, the Gaussian row is the reference)
  if (getenv("DIST_EXP")) {
    const Shape shapes2[] = {{"qkv  ropekv    ", 3 * 4096, 4096, EPI_ROPE_KV}, {"o    residual  ", 4096, 4096, EPI_RESIDUAL},
                             {"gu   swiglu    ", 2 * 11008, 4096, EPI_SWIGLU}, {"down residual  ", 4096, 11008, EPI_RESIDUAL},
                             {"head f32       ", 32000, 4096, EPI_F32}};
    const char* kinds[] = {"gaussian 0.02     ", "student-t df 3    ", "outliers ch30 row8", "log-normal rows   ", "student-t df 2    ", "outliers ch30 ro20"};
    for (const Shape& sh : shapes2) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2;
      const int ntu = gemv_unit_tiles(sh.epi, sh.N), un = sh.N / 16 / ntu;
      const size_t nblk = (size_t)sh.N / 16 * sh.K / 32;
      bf16_t* gw; uint8_t *gc, *ge; uint16_t* gp; int* cnt;
      CK(hipMalloc((void**)&gw, wbytes)); CK(hipMalloc((void**)&gc, nblk * 768)); CK(hipMalloc((void**)&ge, nblk + 8));
      CK(hipMalloc((void**)&gp, (size_t)un * 512)); CK(hipMalloc((void**)&cnt, (size_t)(un + 1) * 4));
      for (int kind = 0; kind < 6; ++kind) {
        hipLaunchKernelGGL(dist_kernel, dim3(4096), dim3(256), 0, 0, gw, sh.N, sh.K, kind, 99u);
        CK(hipMemset(cnt + un, 0, 4));
        CK(launch_compress_weight(gw, sh.N, sh.K, ntu, gc, ge, gp, cnt, cnt + un, 0));
        std::vector<int> hc(un + 1);
        CK(hipMemcpy(hc.data(), cnt, (size_t)(un + 1) * 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> he(nblk);
        CK(hipMemcpy(he.data(), ge, nblk, hipMemcpyDeviceToHost));
        long esc = 0, fl = 0; int mx = 0;
        for (int i = 0; i < un; ++i) { esc += hc[i]; mx = hc[i] > mx ? hc[i] : mx; }
        for (size_t i = 0; i < nblk; ++i) fl += he[i] >> 7;
        GemvArgs b = {};
        b.X = X; b.ldx = sh.K; b.R = R; b.ldr = sh.N; b.C = Y;
        b.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : (sh.epi == EPI_ROPE_KV ? d : sh.N);
        b.M = M; b.N = sh.N; b.K = sh.K; b.Nvalid = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N;
        b.H = H; b.kbeg = ints + 16; b.cur_len = ints; b.cos_tab = cosb; b.sin_tab = sinb;
        b.maxpos = 4096; b.Kc = Kc; b.Vc = Vc; b.c_bs = (long)H * Tmax * 128; b.c_hs = (long)Tmax * 128;
        b.ssq_out = ssq; b.x_xf = 1; b.c_xf = 1;
        if (use_meta) { b.row_meta = row_meta; b.row_cos = row_cos; b.row_sin = row_sin; }
        b.Wp = gw;
        hipError_t err = hipSuccess;
        const float us_plain = time_it([&](int i) { hipError_t e = launch_gemv_ex(b, sh.epi, 0, s); if (e != hipSuccess) err = e; });
        float us_cw = 0.f;
        if (!hc[un]) {
          b.Wp = nullptr; b.Wc = gc; b.emin = ge; b.patch = gp;
          us_cw = time_it([&](int i) { hipError_t e = launch_gemv_ex(b, sh.epi, 0, s); if (e != hipSuccess) err = e; });
        }
        printf("%s %s: escapes %.2e per element, max %3d per unit, flagged blocks %5.1f %%, %s | bf16 stream %6.2f us, 12-bit stream %6.2f us%s\n",
               sh.name, kinds[kind], (double)esc / ((double)sh.N * sh.K), mx, 100.0 * fl / nblk, hc[un] ? "OVERFLOW -> stays bf16" : "compressed",
               us_plain, us_cw, err != hipSuccess ? "  LAUNCH ERROR" : "");
      }
      CK(hipFree(gw)); CK(hipFree(gc)); CK(hipFree(ge)); CK(hipFree(gp)); CK(hipFree(cnt));
    }
    return 0;
  }
  // ---- where does a launch's time go? per-workgroup s_memrealtime stamps (10 ns) of two back-to-back launches
  if (getenv("TS_EXP")) {
    uint8_t* emin;
    uint16_t* patch;
    unsigned long long* ts;
    CK(hipMalloc((void**)&emin, (size_t)8 << 20));
    CK(hipMemset(emin, 56, (size_t)8 << 20));
    CK(hipMalloc((void**)&patch, (size_t)4096 * 256 * 2));
    CK(hipMemset(patch, 0, (size_t)4096 * 256 * 2));
    const int NL = 6;
    CK(hipMalloc((void**)&ts, (size_t)NL * 1024 * 20 * 8));
    for (const Shape& sh : {shapes[0], shapes[1], shapes[3]}) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2;
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.bias = nullptr; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_ROPE_KV ? d : sh.N;
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.ssq_out = ssq; a.x_xf = 1; a.c_xf = 1;
      if (use_meta) { a.row_meta = row_meta; a.row_cos = row_cos; a.row_sin = row_sin; }
      a.emin = emin; a.patch = patch; a.dbg = 4;
      const int grid = sh.N / 16 / (sh.epi == EPI_ROPE_KV ? 3 : 1);
      for (int cw = 0; cw < 2; ++cw) {
        CK(hipMemset(ts, 0, (size_t)NL * 1024 * 20 * 8));
        for (int rep = 0; rep < 3; ++rep)
          for (int i = 0; i < NL; ++i) {
            const size_t off = (((size_t)(i + rep * NL) * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
            a.Wp = (const bf16_t*)((char*)pool + off);
            a.Wc = cw ? (const uint8_t*)pool + off : nullptr;
            a.dbg_ts = ts + (size_t)i * 1024 * 20;
            CK(launch_gemv_ex(a, sh.epi, 0, s));
          }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)NL * 1024 * 20);
        CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long prev_end = 0;
        for (int i = 0; i < NL; ++i) {
          unsigned long long t0min = ~0ull, t0max = 0, t1min = ~0ull, t1max = 0, t2min = ~0ull, t2max = 0, t3min = ~0ull, t3max = 0;
          double sum1 = 0, sum2 = 0, sum3 = 0;
          for (int g = 0; g < grid; ++g) {
            const unsigned long long* q = &h[(size_t)i * 1024 * 20 + (size_t)g * 4];
            t0min = q[0] < t0min ? q[0] : t0min; t0max = q[0] > t0max ? q[0] : t0max;
            t1min = q[1] < t1min ? q[1] : t1min; t1max = q[1] > t1max ? q[1] : t1max;
            t2min = q[2] < t2min ? q[2] : t2min; t2max = q[2] > t2max ? q[2] : t2max;
            t3min = q[3] < t3min ? q[3] : t3min; t3max = q[3] > t3max ? q[3] : t3max;
            sum1 += (double)(q[1] - q[0]); sum2 += (double)(q[2] - q[1]); sum3 += (double)(q[3] - q[2]);
          }
          printf("%s %s launch %d: gap after previous end %6.2f us | entry spread %5.2f | first fragment: mean %5.2f (last wg at %5.2f) | k-loop: mean %5.2f (last done at %5.2f) | epilogue: mean %5.2f | last wg end at %6.2f us\n",
                 sh.name, cw ? "compressed" : "plain     ", i, prev_end ? (double)(long long)(t0min - prev_end) * 0.01 : 0.0, (t0max - t0min) * 0.01,
                 sum1 / grid * 0.01, (t1max - t0min) * 0.01, sum2 / grid * 0.01, (t2max - t0min) * 0.01, sum3 / grid * 0.01, (t3max - t0min) * 0.01);
          prev_end = t3max;
          if (i == NL - 1) {   // per wave: mean / max over the workgroups of (entry -> k-loop done)
            printf("    per-wave k-loop done (mean us over workgroups):");
            for (int wv = 0; wv < 16; ++wv) {
              double sm = 0; int cnt = 0;
              for (int g = 0; g < grid; ++g) { const unsigned long long v = h[(size_t)i * 1024 * 20 + (size_t)grid * 4 + (size_t)g * 16 + wv]; if (v) { sm += (double)v; ++cnt; } }
              if (cnt) printf(" %5.2f", sm / cnt * 0.01);
            }
            printf("\n");
          }
        }
      }
    }
    {   // decode attention, the same stamps
      DecodeAttnArgs a = {};
      a.o_xf = 1;
      a.Q = X; a.q_bs = d; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128; a.O = Y; a.o_bs = d;
      a.part = part; a.kbeg = ints + 16; a.cur_len = ints; a.B = M; a.H = H; a.scale = 0.0883883f;
      const int grid = M * H;
      for (int variant : {0, 14, 17}) {
        CK(hipMemset(ts, 0, (size_t)NL * 1024 * 20 * 8));
        for (int rep = 0; rep < 3; ++rep)
          for (int i = 0; i < NL; ++i) {
            a.Kc = Kc + (size_t)((i + rep) % 8) * kv_elems;
            a.Vc = Vc + (size_t)((i + rep) % 8) * kv_elems;
            a.dbg_ts = ts + (size_t)i * 1024 * 20;
            CK(launch_decode_attention(a, s, variant));
          }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)NL * 1024 * 20);
        CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long prev_end = 0;
        for (int i = 0; i < NL; ++i) {
          unsigned long long t0min = ~0ull, t0max = 0, t2max = 0, t3max = 0;
          double sum1 = 0, sum2 = 0, sum3 = 0;
          for (int g = 0; g < grid; ++g) {
            const unsigned long long* q = &h[(size_t)i * 1024 * 20 + (size_t)g * 4];
            t0min = q[0] < t0min ? q[0] : t0min; t0max = q[0] > t0max ? q[0] : t0max;
            t2max = q[2] > t2max ? q[2] : t2max; t3max = q[3] > t3max ? q[3] : t3max;
            sum1 += (double)(q[1] - q[0]); sum2 += (double)(q[2] - q[1]); sum3 += (double)(q[3] - q[2]);
          }
          if (i >= NL - 2)
            printf("attention v%d launch %d: gap %6.2f us | entry spread %5.2f | lengths known: mean %5.2f | key loop (wave 0): mean %5.2f (last done at %5.2f) | merge + store: mean %5.2f | last wg end at %6.2f us\n",
                   variant, i, prev_end ? (double)(long long)(t0min - prev_end) * 0.01 : 0.0, (t0max - t0min) * 0.01, sum1 / grid * 0.01,
                   sum2 / grid * 0.01, (t2max - t0min) * 0.01, sum3 / grid * 0.01, (t3max - t0min) * 0.01);
          prev_end = t3max;
          if (i == NL - 1) {
            printf("    per-wave key loop done (mean us over workgroups):");
            for (int wv = 0; wv < 16; ++wv) {
              double sm = 0; int cnt = 0;
              for (int g = 0; g < grid; ++g) { const unsigned long long v = h[(size_t)i * 1024 * 20 + (size_t)grid * 4 + (size_t)g * 16 + wv]; if (v) { sm += (double)v; ++cnt; } }
              if (cnt) printf(" %5.2f", sm / cnt * 0.01);
            }
            printf("\n");
          }
        }
      }
    }
    return 0;
  }
  // ---- what bounds the one-unit kernels? the same launch with the activation requests and / or the weight requests dropped by the
  // descriptors' range check (instruction stream, waits and epilogue unchanged)
  if (getenv("DBG_EXP")) {
    uint8_t* emin;
    uint16_t* patch;
    CK(hipMalloc((void**)&emin, (size_t)8 << 20));
    CK(hipMemset(emin, 56, (size_t)8 << 20));
    CK(hipMalloc((void**)&patch, (size_t)4096 * 256 * 2));
    CK(hipMemset(patch, 0, (size_t)4096 * 256 * 2));
    for (const Shape& sh : {shapes[0], shapes[1], shapes[3]}) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2;
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.bias = nullptr; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_ROPE_KV ? d : sh.N;
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.ssq_out = ssq; a.x_xf = 1; a.c_xf = 1;
      if (use_meta) { a.row_meta = row_meta; a.row_cos = row_cos; a.row_sin = row_sin; }
      a.emin = emin; a.patch = patch;
      for (int cw = 0; cw < 2; ++cw)
        for (int dbg = 0; dbg < 4; ++dbg)
          for (int rows : {1, 8, 16}) {
            a.dbg = dbg; a.M = rows;
            hipError_t err = hipSuccess;
            float us = time_it([&](int i) {
              const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
              a.Wp = (const bf16_t*)((char*)pool + off);
              a.Wc = cw ? (const uint8_t*)pool + off : nullptr;
              hipError_t e = launch_gemv_ex(a, sh.epi, 0, s);
              if (e != hipSuccess) err = e;
            });
            printf("%s %s rows %2d %s%s : %7.2f us%s\n", sh.name, cw ? "compressed" : "plain     ", rows, (dbg & 1) ? "no-x " : "     ",
                   (dbg & 2) ? "no-w " : "     ", us, err != hipSuccess ? "  LAUNCH ERROR" : "");
          }
    }
    return 0;
  }
  // ---- compressed weight stream (12-bit blocks, gemv.hip CW kernels): timing on arbitrary compressed bytes, no escapes
  if (getenv("CW_EXP")) {
    uint8_t* emin;
    uint16_t* patch;
    CK(hipMalloc((void**)&emin, (size_t)8 << 20));
    CK(hipMemset(emin, 56, (size_t)8 << 20));          // emin = 112, no escape flag
    CK(hipMalloc((void**)&patch, (size_t)4096 * 256 * 2));
    CK(hipMemset(patch, 0, (size_t)4096 * 256 * 2));
    for (const Shape& sh : shapes) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2, cbytes = wbytes / 4 * 3;
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.bias = nullptr; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : (sh.epi == EPI_ROPE_KV ? d : sh.N);
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.ssq_out = ssq; a.x_xf = 1; a.c_xf = 1;
      if (use_meta) { a.row_meta = row_meta; a.row_cos = row_cos; a.row_sin = row_sin; }
      a.emin = emin; a.patch = patch;
      for (int cw = 0; cw < 2; ++cw) {
        hipError_t err = hipSuccess;
        float us = time_it([&](int i) {
          const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
          a.Wp = (const bf16_t*)((char*)pool + off);
          a.Wc = cw ? (const uint8_t*)pool + off : nullptr;
          hipError_t e = launch_gemv_ex(a, sh.epi, 0, s);
          if (e != hipSuccess) err = e;
        });
        if (err != hipSuccess) { printf("%s cw%d : %s\n", sh.name, cw, hipGetErrorString(err)); (void)hipGetLastError(); continue; }
        printf("%s %s : %7.2f us  (%5.1f MB streamed, %7.1f GB/s physical)\n", sh.name, cw ? "compressed" : "plain     ", us,
               (cw ? cbytes : wbytes) * 1e-6, (cw ? cbytes : wbytes) / us * 1e-3);
      }
      {   // the r2 kernels (branchy k-loop) on the same compressed bytes
        gemv_force_legacy(1);
        hipError_t err = hipSuccess;
        float us = time_it([&](int i) {
          const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
          a.Wp = (const bf16_t*)((char*)pool + off);
          a.Wc = (const uint8_t*)pool + off;
          hipError_t e = launch_gemv_ex(a, sh.epi, 0, s);
          if (e != hipSuccess) err = e;
        });
        gemv_force_legacy(0);
        printf("%s compressed r2 kernel : %7.2f us%s\n", sh.name, us, err != hipSuccess ? "  LAUNCH ERROR" : "");
      }
      for (int variant : {7, 8, 10, 11, 13, 14, 15, 30, 31, 33, 34, 35, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51}) {   // one unit per workgroup: W in {4, 8, 16}, U in {2, 4} (15: W16 U1); 20 + v: two-tile units; 40+: pipelined
        if (variant >= 20 && variant < 40 && sh.epi != EPI_RESIDUAL) continue;
        if (getenv("QUICK") && variant < 40) continue;
        if (variant >= 40) {
          hipError_t err = hipSuccess;
          float us = time_it([&](int i) {
            const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
            a.Wp = (const bf16_t*)((char*)pool + off);
            a.Wc = (const uint8_t*)pool + off;
            hipError_t e = launch_gemv_ex(a, sh.epi, variant, s);
            if (e != hipSuccess) err = e;
          });
          if (err != hipSuccess) { (void)hipGetLastError(); continue; }
          printf("%s compressed pipe W%-2d U%d NB%d : %7.2f us   %7.1f GB/s physical\n", sh.name, variant >= 46 ? 16 : 8, ((variant - 40) % 6) / 3 + 1, (variant - 40) % 3 + 2, us, cbytes / us * 1e-3);
          continue;
        }
        hipError_t err = hipSuccess;
        float us = time_it([&](int i) {
          const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
          a.Wp = (const bf16_t*)((char*)pool + off);
          a.Wc = (const uint8_t*)pool + off;
          hipError_t e = launch_gemv_ex(a, sh.epi, variant, s);
          if (e != hipSuccess) err = e;
        });
        if (err != hipSuccess) { (void)hipGetLastError(); continue; }
        const int vi = variant % 20 - 1;
        printf("%s compressed %sW%-2d U%d : %7.2f us   %7.1f GB/s physical\n", sh.name, variant >= 20 ? "NT2 " : "", 1 << (vi / 3), variant % 20 == 15 ? 1 : 2 << (vi % 3), us, cbytes / us * 1e-3);
      }
      {   // the same launch on REAL compressed data: Gaussian weights through the compressor (escapes on ~5 % of the blocks)
        bf16_t* gw;
        uint8_t *gc, *ge;
        uint16_t* gp;
        int* cnt;
        const int ntu = gemv_unit_tiles(sh.epi, sh.N), un = sh.N / 16 / ntu;
        const size_t nblk = (size_t)sh.N / 16 * sh.K / 32;
        CK(hipMalloc((void**)&gw, wbytes)); CK(hipMalloc((void**)&gc, nblk * 768)); CK(hipMalloc((void**)&ge, nblk + 8));
        CK(hipMalloc((void**)&gp, (size_t)un * 512)); CK(hipMalloc((void**)&cnt, (size_t)(un + 1) * 4));
        hipLaunchKernelGGL(gauss_kernel, dim3(4096), dim3(256), 0, 0, gw, wbytes / 2, 99u);
        CK(hipMemset(cnt + un, 0, 4));
        CK(launch_compress_weight(gw, sh.N, sh.K, ntu, gc, ge, gp, cnt, cnt + un, 0));
        std::vector<int> hc(un + 1);
        CK(hipMemcpy(hc.data(), cnt, (size_t)(un + 1) * 4, hipMemcpyDeviceToHost));
        long esc = 0; int mx = 0;
        for (int i = 0; i < un; ++i) { esc += hc[i]; mx = hc[i] > mx ? hc[i] : mx; }
        GemvArgs b = a;
        b.Wp = nullptr; b.Wc = gc; b.emin = ge; b.patch = gp;
        hipError_t err = hipSuccess;
        float us = time_it([&](int i) { hipError_t e = launch_gemv_ex(b, sh.epi, 0, s); if (e != hipSuccess) err = e; });   // same weights every launch: partly cache-resident
        hipLaunchKernelGGL(clear_flags_kernel, dim3(1024), dim3(256), 0, 0, ge, nblk);   // same bytes and bases, no block flagged
        float us0 = time_it([&](int i) { hipError_t e = launch_gemv_ex(b, sh.epi, 0, s); if (e != hipSuccess) err = e; });
        printf("%s compressed Gaussian weights: %ld escapes (%.2e per element, max %d per unit, overflow %d): %7.2f us; same bytes with no block flagged: %7.2f us%s\n",
               sh.name, esc, (double)esc / ((double)sh.N * sh.K), mx, hc[un], us, us0, err != hipSuccess ? "  LAUNCH ERROR" : "");
        if (sh.epi == EPI_SWIGLU || sh.epi == EPI_F32) {   // persistent configurations on the real data (flags were cleared above: recompress)
          CK(hipMemset(cnt + un, 0, 4));
          CK(launch_compress_weight(gw, sh.N, sh.K, ntu, gc, ge, gp, cnt, cnt + un, 0));
          for (int variant = 40; variant <= 51; ++variant) {   // one unit per workgroup on the pipelined kernel (W, U, NB)
            hipError_t e2 = hipSuccess;
            float t = time_it([&](int i) { hipError_t e = launch_gemv_ex(b, sh.epi, variant, s); if (e != hipSuccess) e2 = e; });
            if (e2 != hipSuccess) { (void)hipGetLastError(); continue; }
            printf("%s Gaussian compressed pipe W%-2d U%d NB%d (one unit per workgroup) : %7.2f us\n", sh.name, variant >= 46 ? 16 : 8, ((variant - 40) % 6) / 3 + 1, (variant - 40) % 3 + 2, t);
          }
          for (int waves : {8, 12, 16})
            for (int u : {1, 2})
              for (int grid : {230, 250, 344, 500}) {
                if (grid > un) continue;
                hipError_t e2 = hipSuccess;
                float t = time_it([&](int i) { hipError_t e = launch_gemv_pers(b, sh.epi, waves, u, grid, s); if (e != hipSuccess) e2 = e; });
                if (e2 != hipSuccess) { (void)hipGetLastError(); continue; }
                printf("%s Gaussian compressed pers W%-2d U%d grid %3d : %7.2f us\n", sh.name, waves, u, grid, t);
              }
        }
        CK(hipFree(gw)); CK(hipFree(gc)); CK(hipFree(ge)); CK(hipFree(gp)); CK(hipFree(cnt));
      }
      if (getenv("QUICK")) continue;
      // persistent sweep on the compressed stream
      const int units = sh.N / 16 / (sh.epi == EPI_RESIDUAL ? 1 : (sh.epi == EPI_ROPE_KV ? 3 : 2));
      for (int waves : {8, 12, 16})
        for (int u : {1, 2})
          for (int grid : {units, (units + 1) / 2, (units + 2) / 3, (units + 3) / 4, 460, 500, 512}) {
            if (grid > 1024 || grid < 128 || grid > units) continue;
            hipError_t err = hipSuccess;
            float us = time_it([&](int i) {
              const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
              a.Wp = (const bf16_t*)((char*)pool + off);
              a.Wc = (const uint8_t*)pool + off;
              hipError_t e = launch_gemv_pers(a, sh.epi, waves, u, grid, s);
              if (e != hipSuccess) err = e;
            });
            if (err != hipSuccess) { (void)hipGetLastError(); continue; }
            printf("%s compressed pers W%-2d U%d grid %4d : %7.2f us  %7.1f GB/s physical\n", sh.name, waves, u, grid, us, cbytes / us * 1e-3);
          }
    }
  }
  // ---- REAL_EXP: every decode GEMV on REAL compressed Gaussian weights that do not fit the Infinity Cache: R copies of the compressed
  // matrix (same bytes, different addresses, > 512 MB in all) rotated launch by launch. What the engine sees; use it to tune.
  if (getenv("REAL_EXP")) {
    for (const Shape& sh : shapes) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2;
      const int ntu = gemv_unit_tiles(sh.epi, sh.N), un = sh.N / 16 / ntu;
      const size_t nblk = (size_t)sh.N / 16 * sh.K / 32, cbytes = nblk * 768, ebytes = (nblk + 255) & ~(size_t)255;
      const int NR = (int)(((size_t)600 << 20) / cbytes) + 1;
      bf16_t* gw;
      uint8_t *gc, *ge;
      uint16_t* gp;
      int* cnt;
      CK(hipMalloc((void**)&gw, wbytes)); CK(hipMalloc((void**)&gc, cbytes * NR)); CK(hipMalloc((void**)&ge, ebytes * NR));
      CK(hipMalloc((void**)&gp, (size_t)un * 512)); CK(hipMalloc((void**)&cnt, (size_t)(un + 1) * 4));
      hipLaunchKernelGGL(gauss_kernel, dim3(4096), dim3(256), 0, 0, gw, wbytes / 2, 99u);
      CK(hipMemset(cnt + un, 0, 4));
      const int grp = getenv("ROWMAJOR") ? 0 : kCwGroup;   // ROWMAJOR=1: the r2 block order
      CK(launch_compress_weight(gw, sh.N, sh.K, ntu, gc, ge, gp, cnt, cnt + un, 0, grp));
      CK(hipDeviceSynchronize());
      for (int r = 1; r < NR; ++r) {
        CK(hipMemcpy(gc + cbytes * r, gc, cbytes, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(ge + ebytes * r, ge, nblk, hipMemcpyDeviceToDevice));
      }
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.bias = nullptr; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : (sh.epi == EPI_ROPE_KV ? d : sh.N);
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.ssq_out = ssq; a.x_xf = 1; a.c_xf = 1; a.patch = gp; a.wc_group = grp;
      if (use_meta) { a.row_meta = row_meta; a.row_cos = row_cos; a.row_sin = row_sin; }
      auto point = [&](int i) { a.Wc = gc + cbytes * (i % NR); a.emin = ge + ebytes * (i % NR); };
      for (int variant : {0, 40, 41, 43, 44, 46, 47, 49, 50}) {
        hipError_t err = hipSuccess;
        float us = time_it([&](int i) { point(i); hipError_t e = launch_gemv_ex(a, sh.epi, variant, s); if (e != hipSuccess) err = e; });
        if (err != hipSuccess) { (void)hipGetLastError(); continue; }
        if (variant == 0) printf("%s REAL default          : %7.2f us  (%d copies of %.1f MB)\n", sh.name, us, NR, cbytes * 1e-6);
        else printf("%s REAL pipe W%-2d U%d NB%d   : %7.2f us\n", sh.name, variant >= 46 ? 16 : 8, ((variant - 40) % 6) / 3 + 1, (variant - 40) % 3 + 2, us);
      }
      if (sh.epi == EPI_SWIGLU || sh.epi == EPI_F32)
        for (int waves : {8, 16})
          for (int u : {1, 2})
            for (int grid : {250, 480, 500}) {
              if (grid > un) continue;
              hipError_t e2 = hipSuccess;
              float t = time_it([&](int i) { point(i); hipError_t e = launch_gemv_pers(a, sh.epi, waves, u, grid, s); if (e != hipSuccess) e2 = e; });
              if (e2 != hipSuccess) { (void)hipGetLastError(); continue; }
              printf("%s REAL pers W%-2d U%d grid %3d : %7.2f us\n", sh.name, waves, u, grid, t);
            }
      CK(hipFree(gw)); CK(hipFree(gc)); CK(hipFree(ge)); CK(hipFree(gp)); CK(hipFree(cnt));
    }
  }
  // ---- persistent GEMV: (waves, U, grid) sweep, checked against the default kernel's output
  if (getenv("PERS_EXP")) {
    for (const Shape& sh : shapes) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2;
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.bias = nullptr; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : (sh.epi == EPI_ROPE_KV ? d : sh.N);
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.ssq_out = ssq;
      const int units = sh.N / 16 / (sh.epi == EPI_RESIDUAL ? 1 : (sh.epi == EPI_ROPE_KV ? 3 : 2));
      // reference output with the default kernel on fixed weights
      const size_t out_bytes = (size_t)16 * 32000 * 4;
      std::vector<char> ref(out_bytes), got(out_bytes);
      a.Wp = pool;
      CK(hipMemset(Y, 0, out_bytes));
      CK(launch_gemv_ex(a, sh.epi, 0, s));
      CK(hipMemcpy(ref.data(), Y, out_bytes, hipMemcpyDeviceToHost));
      for (int waves : {4, 8, 12, 16})
        for (int u : {1, 2})
          for (int div : {1, 2, 3, 4, 5, 6, 8}) {
            int grid = (units + div - 1) / div;
            if (grid > 512 || grid < 64) continue;
            a.Wp = pool;
            CK(hipMemset(Y, 0, out_bytes));
            hipError_t e0 = launch_gemv_pers(a, sh.epi, waves, u, grid, s);
            if (e0 != hipSuccess) { (void)hipGetLastError(); continue; }
            CK(hipMemcpy(got.data(), Y, out_bytes, hipMemcpyDeviceToHost));
            const bool same = memcmp(ref.data(), got.data(), out_bytes) == 0;
            float us = time_it([&](int i) {
              const size_t off = ((size_t)i * wbytes) % (pool_bytes - wbytes);
              a.Wp = (const bf16_t*)((char*)pool + (off & ~(size_t)1023));
              CK(launch_gemv_pers(a, sh.epi, waves, u, grid, s));
            });
            printf("%s pers W%-2d U%d grid %4d (%d units/wg): %7.2f us  %7.1f GB/s  %s\n", sh.name, waves, u, grid, div, us, wbytes / us * 1e-3,
                   same ? "same" : "DIFFERENT");
          }
    }
  }
  // ---- does a prefetch of the head of W (Infinity Cache warm-up during the small norm launch) shorten the GEMV?
  if (getenv("PREFETCH_EXP")) {
    for (const Shape& sh : {shapes[0], shapes[2]}) {
      const size_t wbytes = (size_t)sh.N * sh.K * 2;
      GemvArgs a = {};
      a.X = X; a.ldx = sh.K; a.R = R; a.ldr = sh.N; a.C = Y;
      a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : d;
      a.M = M; a.N = sh.N; a.K = sh.K; a.Nvalid = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N;
      a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.cos_tab = cosb; a.sin_tab = sinb;
      a.maxpos = 4096; a.Kc = Kc; a.Vc = Vc; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      for (size_t mb : {0, 8, 16, 32, 64}) {
        const size_t pf = mb << 20;
        float both = time_it([&](int i) {
          const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
          a.Wp = (const bf16_t*)((char*)pool + off);
          if (pf) hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(512), 0, s, (const u32x4*)((char*)pool + off), pf / 16, (uint32_t*)Y + 64);
          CK(launch_gemv_ex(a, sh.epi, 0, s));
        });
        float rd = 0.f;
        if (pf) rd = time_it([&](int i) {
          const size_t off = (((size_t)i * wbytes) % (pool_bytes - wbytes)) & ~(size_t)1023;
          hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(512), 0, s, (const u32x4*)((char*)pool + off), pf / 16, (uint32_t*)Y + 64);
        });
        printf("%s prefetch %2zu MB : read+gemv %7.2f us, read alone %7.2f us -> gemv part %7.2f us\n", sh.name, mb, both, rd, both - rd);
      }
    }
  }
  // ---- the two small norm launches
  {
    float us = time_it([&](int i) { CK(launch_rmsnorm_part(X, d, ssq, 256, lnw, R, d, M, d, 1e-5f, s)); });
    printf("rmsnorm_part             : %7.2f us\n", us);
    us = time_it([&](int i) { CK(launch_rmsnorm(X, d, lnw, R, d, M, d, 1e-5f, s)); });
    printf("rmsnorm (self)           : %7.2f us\n", us);
  }
  // ---- decode attention
  {
    DecodeAttnArgs a = {};
    a.o_xf = 1;
    a.Q = X; a.q_bs = d; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128; a.O = Y; a.o_bs = d;
    a.part = part; a.kbeg = ints + 16; a.cur_len = ints; a.B = M; a.H = H; a.scale = 0.0883883f;
    const size_t bytes = (size_t)M * H * ctx * 512;
    for (int variant : {0, 100, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18}) {   // 0 = default (pipelined 16 x 1), 100 = r2 kernel, 12..18 pipelined (waves, U)
      if (getenv("QUICK") && variant >= 1 && variant <= 11) continue;
      float us = time_it([&](int i) {
        a.Kc = Kc + (size_t)(i % 8) * kv_elems;
        a.Vc = Vc + (size_t)(i % 8) * kv_elems;
        CK(launch_decode_attention(a, s, variant));
      });
      printf("decode attention B=%d ctx=%d v%d : %7.2f us  %7.1f GB/s\n", M, ctx, variant, us, bytes / us * 1e-3);
    }
  }
  // ---- one full layer sequence (7 launches), distinct weights per call
  {
    const size_t layer_bytes = ((size_t)4 * d * d + (size_t)3 * d * ff) * 2;
    auto layer = [&](int i) {
      char* base = (char*)pool + ((size_t)(i % 6) * layer_bytes);
      GemvArgs a = {};
      a.bias = nullptr; a.M = M; a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.x_xf = 1; a.c_xf = 1;
      a.cos_tab = cosb; a.sin_tab = sinb; a.maxpos = 4096; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.Kc = Kc + (size_t)(i % 8) * kv_elems; a.Vc = Vc + (size_t)(i % 8) * kv_elems;
      bf16_t* xn = R + (size_t)8 * ff;   // scratch rows
      CK(launch_rmsnorm_part(X, d, ssq, 256, lnw, xn, d, M, d, 1e-5f, s, 1));
      a.X = xn; a.ldx = d; a.Wp = (const bf16_t*)base; a.C = Y; a.ldc = d; a.N = 3 * d; a.K = d; a.Nvalid = 3 * d; a.R = nullptr;
      CK(launch_gemv_ex(a, EPI_ROPE_KV, 0, s));
      DecodeAttnArgs t = {};
      t.o_xf = 1;
      t.Q = Y; t.q_bs = d; t.Kc = a.Kc; t.Vc = a.Vc; t.c_bs = a.c_bs; t.c_hs = a.c_hs; t.O = R; t.o_bs = d;
      t.part = part; t.kbeg = ints + 16; t.cur_len = ints; t.B = M; t.H = H; t.scale = 0.0883883f;
      CK(launch_decode_attention(t, s));
      a.X = R; a.Wp = (const bf16_t*)(base + (size_t)3 * d * d * 2); a.C = X; a.R = X; a.ldr = d; a.ldc = d; a.N = d; a.Nvalid = d;
      a.ssq_out = ssq;
      CK(launch_gemv_ex(a, EPI_RESIDUAL, 0, s));
      CK(launch_rmsnorm_part(X, d, ssq, 256, lnw, xn, d, M, d, 1e-5f, s, 1));
      a.X = xn; a.Wp = (const bf16_t*)(base + (size_t)4 * d * d * 2); a.C = Y; a.ldc = ff; a.N = 2 * ff; a.Nvalid = ff; a.R = nullptr;
      a.ssq_out = nullptr;
      CK(launch_gemv_ex(a, EPI_SWIGLU, 0, s));
      a.X = Y; a.ldx = ff; a.K = ff; a.Wp = (const bf16_t*)(base + ((size_t)4 * d * d + (size_t)2 * d * ff) * 2); a.C = X; a.R = X;
      a.ldr = d; a.ldc = d; a.N = d; a.Nvalid = d; a.ssq_out = ssq;
      CK(launch_gemv_ex(a, EPI_RESIDUAL, 0, s));
    };
    // the same layer with both RMSNorms folded into their projections (5 launches): raw rows in, row scale in the epilogue
    // OVERLAP_EXP: consecutive launches alternate between TWO streams with no dependency between the streams (launch j and j + 2
    // stay ordered, j and j + 1 may overlap; the results are garbage, only the timing means something): an upper bound on what
    // letting launch j + 1 start its weight stream under launch j's tail could buy (it would need a device-side hand-off:
    // tools/experiments/r3_device_handoff.patch, measured 2.8x SLOWER than one stream).
    hipStream_t ov[2] = {s, s};
    int ov_cnt = 0;
    auto nexts = [&]() -> hipStream_t { return ov[(ov_cnt++) & 1]; };
    auto layer5 = [&](int i) {
      char* base = (char*)pool + ((size_t)(i % 6) * layer_bytes);
      GemvArgs a = {};
      a.bias = nullptr; a.M = M; a.H = H; a.kbeg = ints + 16; a.cur_len = ints; a.x_xf = 1; a.c_xf = 1;
      a.cos_tab = cosb; a.sin_tab = sinb; a.maxpos = 4096; a.c_bs = (long)H * Tmax * 128; a.c_hs = (long)Tmax * 128;
      a.Kc = Kc + (size_t)(i % 8) * kv_elems; a.Vc = Vc + (size_t)(i % 8) * kv_elems;
      a.n_part = d / 16; a.norm_d = d; a.norm_eps = 1e-5f;
      bf16_t* hxf = R + (size_t)8 * ff;   // fragment-ordered hidden rows
      a.X = hxf; a.ldx = d; a.Wp = (const bf16_t*)base; a.C = Y; a.ldc = d; a.N = 3 * d; a.K = d; a.Nvalid = 3 * d; a.R = nullptr;
      a.ssq_in = ssq;
      CK(launch_gemv_ex(a, EPI_ROPE_KV, 0, nexts()));
      DecodeAttnArgs t = {};
      t.o_xf = 1;
      t.Q = Y; t.q_bs = d; t.Kc = a.Kc; t.Vc = a.Vc; t.c_bs = a.c_bs; t.c_hs = a.c_hs; t.O = R; t.o_bs = d;
      t.part = part; t.kbeg = ints + 16; t.cur_len = ints; t.B = M; t.H = H; t.scale = 0.0883883f;
      CK(launch_decode_attention(t, nexts()));
      a.X = R; a.Wp = (const bf16_t*)(base + (size_t)3 * d * d * 2); a.C = X; a.R = X; a.ldr = d; a.ldc = d; a.N = d; a.Nvalid = d;
      a.ssq_in = nullptr; a.ssq_out = ssq; a.C_xf = hxf;
      CK(launch_gemv_ex(a, EPI_RESIDUAL, 0, nexts()));
      a.X = hxf; a.Wp = (const bf16_t*)(base + (size_t)4 * d * d * 2); a.C = Y; a.ldc = ff; a.N = 2 * ff; a.Nvalid = ff; a.R = nullptr;
      a.ssq_out = nullptr; a.C_xf = nullptr; a.ssq_in = ssq;
      CK(launch_gemv_ex(a, EPI_SWIGLU, 0, nexts()));
      a.X = Y; a.ldx = ff; a.K = ff; a.Wp = (const bf16_t*)(base + ((size_t)4 * d * d + (size_t)2 * d * ff) * 2); a.C = X; a.R = X;
      a.ldr = d; a.ldc = d; a.N = d; a.Nvalid = d; a.ssq_in = nullptr; a.ssq_out = ssq; a.C_xf = hxf;
      CK(launch_gemv_ex(a, EPI_RESIDUAL, 0, nexts()));
    };
    if (getenv("OVERLAP_EXP")) {
      CK(hipStreamCreate(&ov[0]));
      CK(hipStreamCreate(&ov[1]));
      hipEvent_t j0, j1;
      CK(hipEventCreate(&j0));
      CK(hipEventCreate(&j1));
      for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 8; ++i) layer5(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, ov[0]));
        CK(hipEventRecord(j0, ov[0]));
        CK(hipStreamWaitEvent(ov[1], j0, 0));
        for (int i = 0; i < 64; ++i) layer5(i);
        CK(hipEventRecord(j1, ov[1]));
        CK(hipStreamWaitEvent(ov[0], j1, 0));
        CK(hipEventRecord(e1, ov[0]));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("full layer (5 launches) alternating over two unordered streams : %7.2f us per layer\n", ms * 1e3f / 64);
      }
      ov[0] = ov[1] = s;
      CK(hipDeviceSynchronize());
    }
    float us5 = time_it(layer5);
    float us = time_it(layer);
    us5 = 0.5f * (us5 + time_it(layer5));
    if (getenv("GRAPH_EXP")) {
      // the same 32-layer step as ONE hipGraph launch: does replaying pre-built packets shorten the gaps between kernels?
      hipStream_t cs;
      CK(hipStreamCreate(&cs));
      hipStream_t saved = s;
      s = cs;
      hipGraph_t graph; hipGraphExec_t exec;
      CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
      for (int i = 0; i < 32; ++i) layer(i);
      CK(hipStreamEndCapture(cs, &graph));
      CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, cs));
      CK(hipEventRecord(e0, cs));
      for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(exec, cs));
      CK(hipEventRecord(e1, cs));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("32-layer step as one hipGraph : %7.2f us per layer\n", ms * 1e3f / (20 * 32));
      s = saved;
    }
    const size_t bytes = layer_bytes + (size_t)M * H * ctx * 512;
    printf("full layer (7 launches)      : %7.2f us  %7.1f GB/s  -> %.2f ms / 32-layer step\n", us, bytes / us * 1e-3, us * 32e-3);
    printf("full layer (5 launches, norms folded) : %7.2f us  %7.1f GB/s  -> %.2f ms / 32-layer step\n", us5, bytes / us5 * 1e-3, us5 * 32e-3);
  }
  return 0;
}

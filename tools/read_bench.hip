// HBM streaming-read ceiling by launch shape / address pattern (GPU box only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// pattern 0: chip-wide grid-stride in 1-KiB wave pieces: piece index = it*GW + gw  (GW = total waves)
// pattern 1: per-workgroup contiguous region, waves interleaved inside it (piece = wg*per_wg + it*W + w)
// pattern 2: per-wave contiguous region (piece = gw*per_wave + it)
template <int U>
__global__ void read_kernel(const u32x4* __restrict__ p, size_t npieces, int pattern, uint32_t* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const size_t GW = (size_t)gridDim.x * W, gw = (size_t)blockIdx.x * W + w;
  size_t start, step, count;
  if (pattern == 0) { start = gw; step = GW; count = (npieces - gw + GW - 1) / GW; }
  else if (pattern == 1) { size_t per = npieces / gridDim.x; start = blockIdx.x * per + w; step = W; count = (per - w + W - 1) / W; }
  else { size_t per = npieces / GW; start = gw * per; step = 1; count = per; }
  u32x4 acc = {0u, 0u, 0u, 0u};
  size_t i = 0;
  for (; i + U <= count; i += U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (start + (i + u) * step) * 64 + lane);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  for (; i < count; ++i) acc ^= __builtin_nontemporal_load(p + (start + i * step) * 64 + lane);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

int main() {
  const size_t pool_bytes = (size_t)3 << 30;
  char* pool; uint32_t* out;
  CK(hipMalloc((void**)&pool, pool_bytes)); CK(hipMalloc((void**)&out, 64));
  CK(hipMemset(pool, 1, pool_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 30;
  for (size_t mb : {32, 96, 172}) {
    const size_t bytes = mb << 20, np = bytes / 1024;
    for (int pattern = 0; pattern < 3; ++pattern)
      for (int threads : {256, 512, 1024})
        for (int grid : {256, 512, 1024})
          for (int U : {2, 4, 8, 16}) {
            auto launch = [&](int i) {
              const size_t off = (((size_t)i * bytes) % (pool_bytes - bytes)) & ~(size_t)4095;
              const u32x4* p = (const u32x4*)(pool + off);
              if (U == 2) hipLaunchKernelGGL(read_kernel<2>, dim3(grid), dim3(threads), 0, 0, p, np, pattern, out);
              if (U == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(threads), 0, 0, p, np, pattern, out);
              if (U == 8) hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(threads), 0, 0, p, np, pattern, out);
              if (U == 16) hipLaunchKernelGGL(read_kernel<16>, dim3(grid), dim3(threads), 0, 0, p, np, pattern, out);
            };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch(i + 3);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float us = ms * 1e3f / iters;
            printf("%3zuMB pat%d thr%4d grid%4d U%2d : %7.2f us %7.1f GB/s\n", mb, pattern, threads, grid, U, us, bytes / us * 1e-3);
          }
  }
  return 0;
}

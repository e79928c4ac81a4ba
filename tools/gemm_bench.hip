// Micro-benchmark of the large-M GEMM kernels on the prefill / ViT shapes (GPU box only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../oracle_amd/csrc/common.h"
#include "../oracle_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (uint16_t)(((h & 1) << 15) | (0x3C00 + (h >> 8 & 0xFF)));
  }
}
struct Shape { const char* name; int M, N, K, epi; };
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8;
  const int Mp = B * 739, Mv = B * 3 * 577;
  const Shape shapes[] = {
      {"prefill qkv      ", Mp, 12288, 4096, EPI_NONE}, {"prefill o  +res  ", Mp, 4096, 4096, EPI_RESIDUAL},
      {"prefill gu swiglu", Mp, 22016, 4096, EPI_SWIGLU}, {"prefill down+res ", Mp, 4096, 11008, EPI_RESIDUAL},
      {"vit qkv          ", Mv, 3072, 1024, EPI_NONE}, {"vit o +res       ", Mv, 1024, 1024, EPI_RESIDUAL},
      {"vit fc1 quickgelu", Mv, 4096, 1024, EPI_QUICKGELU}, {"vit fc2 +res     ", Mv, 1024, 4096, EPI_RESIDUAL},
      {"square 4096      ", 4096, 4096, 4096, EPI_NONE}, {"square 8192      ", 8192, 8192, 8192, EPI_NONE},   // the guide's template shapes
  };
  bf16_t *X, *W, *R, *C, *bias;
  const size_t w_elems = getenv("BIGK") ? (size_t)22016 * (16384 + 4096) : (size_t)22016 * 4096 * 2;
  CK(hipMalloc((void**)&X, (size_t)Mv * 11008 * 2)); CK(hipMalloc((void**)&W, w_elems * 2));   // X >= 8192 x 8192, W >= 2 x 8192 x 8192
  CK(hipMalloc((void**)&R, (size_t)Mv * 4096 * 2)); CK(hipMalloc((void**)&C, (size_t)Mv * 12288 * 2)); CK(hipMalloc((void**)&bias, 65536));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, (size_t)Mv * 11008, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, w_elems, 2u);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, R, (size_t)Mv * 4096, 3u);
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, bias, 32768, 4u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {   // clocks ramp over the first tenths of a second of load: warm the chip before the first timed shape
    GemmArgs a; a.X = X; a.ldx = 4096; a.Wp = W; a.bias = bias; a.R = R; a.ldr = 4096; a.C = C; a.ldc = 4096; a.M = Mp; a.N = 4096; a.K = 4096; a.Nvalid = 4096;
    for (int i = 0; i < 1500; ++i) CK(launch_gemm(a, EPI_NONE, 0));
    CK(hipDeviceSynchronize());
  }
  if (getenv("BIGK")) {
    // Is the prefill gate/up launch's fabric traffic (PMC FETCH_SIZE 2.04 GB against 0.36 GB of algorithmic bytes: every tile re-reads
    // its operand panels past the 4-MiB L2) a cost? Its operands (228 MB) fit the 256-MiB Infinity Cache; the same launch with K x 4
    // (915 MB of operands, 3.6 x the cache) cannot be served from it. Equal TFLOP/s = the re-reads are not what bounds the kernel.
    const Shape big[] = {{"prefill gu swiglu      ", Mp, 22016, 4096, EPI_SWIGLU}, {"prefill gu swiglu K x 4", Mp, 22016, 16384, EPI_SWIGLU}};
    for (const Shape& sh : big) {
      GemmArgs a; a.X = X; a.ldx = sh.K; a.Wp = W; a.bias = bias; a.R = R; a.ldr = sh.N; a.C = C;
      a.ldc = sh.N / 2; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.Nvalid = a.ldc;
      gemm_force_kernel(0);
      for (int i = 0; i < 4; ++i) CK(launch_gemm(a, sh.epi, 0));
      CK(hipEventRecord(e0, 0));
      const int iters = 10;
      for (int i = 0; i < iters; ++i) { a.Wp = W + (size_t)(i & 1) * 22016 * 4096; CK(launch_gemm(a, sh.epi, 0)); }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("%s M=%5d N=%5d K=%5d operands %4.0f MB : %8.1f us  %7.1f TFLOP/s\n", sh.name, sh.M, sh.N, sh.K,
             ((double)sh.M * sh.K + (double)sh.N * sh.K) * 2e-6, us, 2.0 * sh.M * sh.N * sh.K / us * 1e-6);
    }
    return 0;
  }
  for (const Shape& sh : shapes)
    for (int force = 0; force <= 3; ++force) {
      GemmArgs a; a.X = X; a.ldx = sh.K; a.Wp = W; a.bias = bias; a.R = R; a.ldr = sh.N; a.C = C;
      a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.Nvalid = a.ldc;
      gemm_force_kernel(force);
      for (int i = 0; i < 6; ++i) CK(launch_gemm(a, sh.epi, 0));
      CK(hipEventRecord(e0, 0));
      const int iters = 10;
      for (int i = 0; i < iters; ++i) { a.Wp = W + (size_t)(i & 1) * 22016 * 4096; CK(launch_gemm(a, sh.epi, 0)); }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("%s M=%5d N=%5d K=%5d %s : %8.1f us  %7.1f TFLOP/s\n", sh.name, sh.M, sh.N, sh.K,
             force == 0 ? "auto   " : (force == 1 ? "tile128" : (force == 2 ? "dma256 " : "dma128 ")), us, 2.0 * sh.M * sh.N * sh.K / us * 1e-6);
    }
  if (getenv("TALL_SWEEP")) {
    for (const Shape& sh : shapes) {
      for (int tall = (sh.M + 255) / 256; tall >= 0; tall -= 1) {
        if (tall < (sh.M + 255) / 256 - 10 && tall % 8) continue;
        GemmArgs a; a.X = X; a.ldx = sh.K; a.Wp = W; a.bias = bias; a.R = R; a.ldr = sh.N; a.C = C;
        a.ldc = sh.epi == EPI_SWIGLU ? sh.N / 2 : sh.N; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.Nvalid = a.ldc;
        gemm_force_kernel(100 + tall);
        for (int i = 0; i < 3; ++i) CK(launch_gemm(a, sh.epi, 0));
        CK(hipEventRecord(e0, 0));
        const int iters = 10;
        for (int i = 0; i < iters; ++i) { a.Wp = W + (size_t)(i & 1) * 22016 * 4096; CK(launch_gemm(a, sh.epi, 0)); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s tall %2d : %8.1f us\n", sh.name, tall, ms * 1e3 / iters);
      }
    }
  }
  return 0;
}

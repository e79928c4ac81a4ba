// Shared device helpers for the ORacle MI355X (gfx950) hot-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // storage type for bf16 in global/LDS memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
// round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
// value of f after a round trip through bf16 -- every point where the reference (torch eager, bf16
// tensors) materialises a tensor is one such rounding.
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// two values through bf16 with one v_cvt_pk_bf16_f32
__device__ __forceinline__ void rbf2(float a, float b, float& ra, float& rb) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  const uint32_t u = __builtin_bit_cast(uint32_t, h);
  ra = __uint_as_float(u << 16);
  rb = __uint_as_float(u & 0xFFFF0000u);
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float lo_bf(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Packed weight layout ("MFMA fragment order"), used by every GEMM/GEMV in the library.
// A row-major weight W[N][K] (nn.Linear: out x in) is stored as 1 KiB blocks, block (nt, kt) holding
// rows 16nt..16nt+15, columns 32kt..32kt+31 in the order a wave's lanes consume them as the A/B operand
// of v_mfma_f32_16x16x32_bf16: lane l, element j  <-  W[16nt + (l & 15)][32kt + 8(l >> 4) + j].
// One wave-wide 16-byte load of a block is therefore a fully contiguous 1 KiB read, and the same
// block copied linearly into LDS is read back conflict-free with ds_read_b128 at lane*16.
__host__ __device__ __forceinline__ size_t packed_block_elems() { return 512; }

// "Fragment-ordered" activation layout of the decode path (<= 16 rows): element (row b, input k) lives at
// ((k / 8) * 16 + b) * 8 + k % 8, i.e. [k-chunk of 8][16 rows][8]. The MFMA B fragment of a k-tile (lane = (g, row): 8
// elements at k = 32 kt + 8 g) is then ONE contiguous 1-KiB wave read (8 whole 128-B lines) instead of 16 half lines of
// 16 different rows; every GEMV workgroup re-reads the whole activation through L2, so this halves its request count.
__host__ __device__ __forceinline__ size_t xf_off(int b, int k) { return ((size_t)(k >> 3) * 16 + b) * 8 + (k & 7); }
// more than 16 rows: one such 16-row image per group of 16 rows, images row_len * 16 elements apart
__host__ __device__ __forceinline__ size_t xf_off_rows(int b, int k, int row_len) {
  return (size_t)(b >> 4) * 16 * row_len + xf_off(b & 15, k);
}

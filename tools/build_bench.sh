#!/bin/bash
# builds the GPU-box micro-benchmarks (not part of the product)
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEMV_ALL_VARIANTS -o tools/decode_bench tools/decode_bench.hip \
  oracle_amd/csrc/gemv.hip oracle_amd/csrc/attn.hip oracle_amd/csrc/norm.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_bench tools/gemm_bench.hip oracle_amd/csrc/gemm.hip

// common.h -- shared definitions of libgpcore (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace gpc {

constexpr int TILE = 128;  // block tile of every dense stage; all padded sizes are multiples
constexpr int WAVE = 64;

inline int pad_tile(int n) { return ((n + TILE - 1) / TILE) * TILE; }

// ---- MFMA 16x16x4 traits (one A element and one B element per lane) ---------------
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D: 4 values per lane, column = l & 15, row = row_of(l, r)  (f64 and f32 differ)
template <typename T>
struct MM;

template <>
struct MM<double> {
  using acc_t = double __attribute__((ext_vector_type(4)));
  using vec_t = double __attribute__((ext_vector_type(2)));  // 16-byte global/LDS vector
  static constexpr int VEC = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
#if defined(GPC_MFMA_AGPR)
    asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
#elif defined(GPC_MFMA_VASM)
    asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#endif
  }
  static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
};

template <>
struct MM<float> {
  using acc_t = float __attribute__((ext_vector_type(4)));
  using vec_t = float __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// wave-level sum (all lanes get the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-level sum for 256-thread blocks; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[w] = v;
  __syncthreads();
  return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

}  // namespace gpc

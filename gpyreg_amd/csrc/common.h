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

// `info` of a factorization: 0 = done, k > 0 = the pivot of row k was not positive (LAPACK's convention; the host
// retries with more jitter, gaussian_process.py:2413-2421), bit 30 = a hand-off inside a leaf timed out (leaf.h) --
// an internal error, reported as such and never retried
constexpr int LEAF_TIMEOUT = 1 << 30;

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

// raw buffer descriptor based at `base` (no bounds: 2 GiB range, 32-bit byte offsets): loads through it take a
// scalar descriptor + one VGPR of offset instead of a 64-bit vector address
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

// ---- persistent launches: tile queues and CU reservation ---------------------------------------
constexpr int NQ = 8;  // tile queues of a persistent GEMM launch (one per XCD, gemm.h)
// Device counters of one persistent launch: NQ tile queues, then the two counters of the CU reservation
constexpr int CTR_STARTED = NQ, CTR_SURVIVORS = NQ + 1, CTR_STRIDE = NQ + 4;

// CU reservation without CU-masked queues.  `tbl` (64 entries, [XCC_ID][SE_ID] -> bit mask over CU_ID; built by the
// host from a probe of THIS device's shader-engine / CU numbering, gpcore.hip: probe_cu_map) marks the CUs a launch
// is to stay off.  A block that finds itself on a marked CU returns at once: the grid is oversized by the caller,
// the surplus drains through the marked CUs in microseconds, and those CUs stay EMPTY for the whole launch -- the
// latency-bound kernels of another stream (128 x 128 leaves, deep-level products) run there at full speed while
// this launch streams on the other CUs.
// The work must get done wherever the dispatcher puts the blocks: every block counts itself in CTR_STARTED, the
// ones that stay also in CTR_SURVIVORS (before), and the block that finds itself the LAST to start with no survivor
// so far stays whatever CU it is on.  So at least one block always serves the queues (tests: a table that marks
// every CU), and no assumption about placement or about the numbering is needed for correctness.
// Returns true when the calling block is to return.  Block-uniform; contains a block barrier.
__device__ __forceinline__ bool cu_reserve_bail(const unsigned short* __restrict__ tbl, int* __restrict__ ctr) {
  __shared__ int bail_sh;
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 0xf, se = (hw >> 13) & 0x7;
    int bail = (tbl[(xcc & 7) * 8 + se] >> cu) & 1;
    if (!bail) {
      __hip_atomic_fetch_add(ctr + CTR_SURVIVORS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(ctr + CTR_STARTED, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int before = __hip_atomic_fetch_add(ctr + CTR_STARTED, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (int)gridDim.x - 1 &&
          __hip_atomic_load(ctr + CTR_SURVIVORS, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0)
        bail = 0;
    }
    bail_sh = bail;
  }
  __syncthreads();
  return bail_sh != 0;
}

}  // namespace gpc

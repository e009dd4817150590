// Can a chain of small "leaf-like" kernels (8 blocks, ~200 VGPRs, little LDS) make progress while a
// bulk kernel occupies the chip, if the bulk kernel leaves k half-empty CUs (grid = 512 - k blocks of
// 72 KB LDS / 256 VGPRs, two per CU)?  diagnostic for the sample-group overlap design
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
// bulk: MFMA loop, 256 VGPRs (launch bound) and 72 KB dynamic LDS -> two blocks per CU
__global__ __launch_bounds__(256, 2) void bulk(double* out, int iters) {
  extern __shared__ double lds[];
  d4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  lds[threadIdx.x] = a;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b + lds[(it + i) & 255], c[i], 0, 0, 0);
  }
  asm volatile("" ::: "v255");  // claim the whole 256-VGPR budget, like the GEMM
  double s = 0;
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
// leaf-like: 8 blocks x 256 threads, a dependent chain of ~80 us, many live registers
__global__ __launch_bounds__(256) void leafish(double* buf, int spin) {
  double r[48];
  for (int i = 0; i < 48; ++i) r[i] = buf[(blockIdx.x * 256 + threadIdx.x) * 64 + i];
  for (int s = 0; s < spin; ++s) {
#pragma unroll
    for (int i = 0; i < 48; ++i) r[i] = r[i] * 0.999 + r[(i + 1) % 48] * 1e-3;
    __syncthreads();
  }
  for (int i = 0; i < 48; ++i) buf[(blockIdx.x * 256 + threadIdx.x) * 64 + i] = r[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  double *o, *b; hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&b, 8 * 256 * 64 * 8); hipMemset(b, 0, 8 * 256 * 64 * 8);
  hipFuncSetAttribute((const void*)bulk, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  const int chain = 100, spin = 600;
  // calibrate
  hipLaunchKernelGGL(leafish, dim3(8), dim3(256), 0, s2, b, spin); hipStreamSynchronize(s2);
  double t0 = now();
  for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(leafish, dim3(8), dim3(256), 0, s2, b, spin);
  hipStreamSynchronize(s2);
  double t_alone = now() - t0;
  printf("chain of %d leaf-like kernels alone: %.2f ms (%.1f us each)\n", chain, t_alone, t_alone * 1e3 / chain);
  for (int spare : {0, 8, 16, 32, 64, 256}) {
    const int grid = 512 - spare, iters = 14000;
    hipLaunchKernelGGL(bulk, dim3(grid), dim3(256), 72 * 1024, s1, o, 10); hipStreamSynchronize(s1);
    t0 = now();
    hipLaunchKernelGGL(bulk, dim3(grid), dim3(256), 72 * 1024, s1, o, iters); hipStreamSynchronize(s1);
    double t_bulk_alone = now() - t0;
    t0 = now();
    hipLaunchKernelGGL(bulk, dim3(grid), dim3(256), 72 * 1024, s1, o, iters);
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(leafish, dim3(8), dim3(256), 0, s2, b, spin);
    hipStreamSynchronize(s2);
    double t_chain = now() - t0;
    hipStreamSynchronize(s1);
    double t_both = now() - t0;
    printf("bulk grid %3d (spare half-CUs %3d): bulk alone %.2f ms; with chain: chain done at %.2f ms, all done at %.2f ms\n", grid, spare,
           t_bulk_alone, t_chain, t_both);
  }
  return 0;
}

// fp64 MFMA issue cadence of a register-resident loop (inline asm, 16 independent accumulators,
// 4 A x 4 B fragments like the GEMM's inner step) vs waves per SIMD and active CUs.  diagnostic
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define M(c, a, b) "v_mfma_f64_16x16x4_f64 %" #c ", %" #a ", %" #b ", %" #c "\n"
template <int VARIANT>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  d4 c0{}, c1{}, c2{}, c3{}, c4{}, c5{}, c6{}, c7{}, c8{}, c9{}, c10{}, c11{}, c12{}, c13{}, c14{}, c15{};
  double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double b0 = seed - threadIdx.x * 1e-3, b1 = b0 - 1, b2 = b0 - 2, b3 = b0 - 3;
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 0)
      asm volatile(M(0, 16, 20) M(1, 16, 21) M(2, 16, 22) M(3, 16, 23) M(4, 17, 20) M(5, 17, 21) M(6, 17, 22) M(7, 17, 23)
                   M(8, 18, 20) M(9, 18, 21) M(10, 18, 22) M(11, 18, 23) M(12, 19, 20) M(13, 19, 21) M(14, 19, 22) M(15, 19, 23)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(c8), "+v"(c9),
                     "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13), "+v"(c14), "+v"(c15)
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    else  // same with a scalar instruction between MFMAs
      asm volatile(M(0, 16, 20) "s_nop 0\n" M(1, 16, 21) "s_nop 0\n" M(2, 16, 22) "s_nop 0\n" M(3, 16, 23) "s_nop 0\n"
                   M(4, 17, 20) "s_nop 0\n" M(5, 17, 21) "s_nop 0\n" M(6, 17, 22) "s_nop 0\n" M(7, 17, 23) "s_nop 0\n"
                   M(8, 18, 20) "s_nop 0\n" M(9, 18, 21) "s_nop 0\n" M(10, 18, 22) "s_nop 0\n" M(11, 18, 23) "s_nop 0\n"
                   M(12, 19, 20) "s_nop 0\n" M(13, 19, 21) "s_nop 0\n" M(14, 19, 22) "s_nop 0\n" M(15, 19, 23)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(c8), "+v"(c9),
                     "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13), "+v"(c14), "+v"(c15)
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
  }
  d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int V>
void sweep(double* d, const char* tag) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000;
  for (int wgs : {1, 64, 256, 512, 768, 1024}) {
    hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(256), 0, 0, d, 100, 1.2345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.2345);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_wave = 16.0 * iters;
    int waves_per_simd = wgs <= 256 ? 1 : (wgs + 255) / 256;
    double ns = ms * 1e6 / (mfma_per_wave * waves_per_simd);
    double tf = (double)wgs * 4 * mfma_per_wave * 2048 / (ms * 1e-3) / 1e12;
    printf("%s WGs %5d (%d waves/SIMD): %8.3f ms  %6.1f ns per MFMA per SIMD (%5.1f cycles at 2.4 GHz)  %6.2f TFLOP/s\n", tag, wgs,
           waves_per_simd, ms, ns, ns * 2.4, tf);
  }
}
int main() {
  double* d; hipMalloc(&d, 4096 * 256 * 8);
  sweep<0>(d, "back-to-back ");
  sweep<1>(d, "s_nop between");
  return 0;
}

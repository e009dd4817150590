// seam_probe: what does a dependent GEMM -> GEMM seam cost on this box as a KERNEL BOUNDARY, and what as an in-kernel
// hand-off?  ("a subtree kernel with workgroup-level release/acquire flags between tiles
// instead of kernel boundaries".)  The chain is the deep levels' own shape: step s computes C_{s+1} = C_s * B with the
// library's 64-tile kernel (gemm.h: gemm_tile<double, false, true, 64, 4>), n x n x n with n = 128 ... 512, i.e.
// 4 ... 64 workgroups per step, every tile of a step reading the whole previous result.
//   A  one launch per step (what plan.h does today)
//   B  ONE launch, one resident workgroup per tile; after its tile a workgroup publishes (write-through sc1 stores,
//      s_waitcnt vmcnt(0), agent-scope counter add) and waits until all tiles of the step have arrived (relaxed sc1
//      poll + s_sleep), operands loaded with sc1 loads -- the forms MI355X_MICROARCH.md prescribes
//   C  as B with plain stores + an agent-scope release fence before the counter and an acquire fence after the wait
// Results of A, B and C are compared bit for bit.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/seam_probe.hip -o seam_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../gpyreg_amd/csrc/gemm.h"

using namespace gpc;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int MODE>  // 1: sc1 stores / loads + drained counter; 2: plain stores + release / acquire fences
__global__ __launch_bounds__(256, 2) void chain_kernel(GemmArgs g, double* buf0, double* buf1, int steps, int* ctr,
                                                       int* timed_out) {
  __shared__ __attribute__((aligned(16))) double smem[4 * opsz_of<double>(64)];
  __shared__ int give_up;
  const int tiles = gridDim.x;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    g.A = (s & 1) ? buf1 : buf0;
    g.C = (s & 1) ? buf0 : buf1;
    gemm_tile<double, false, true, 64, 4, MODE == 1 ? 1 : 0>(g, blockIdx.x, 0, smem);
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int n = 0;
      while (__hip_atomic_load(ctr + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tiles) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > (1 << 22)) {  // bounded: a missed arrival ends the run instead of hanging the queue
          give_up = 1;
          atomicExch(timed_out, 1);
          break;
        }
      }
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (give_up) return;
  }
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%6s %6s | %12s %12s %12s | per step: launches, in-kernel sc1, in-kernel fences (us); equal bits\n", "n", "tiles",
         "A launches", "B sc1", "C fences");
  for (int n : {128, 256, 384, 512}) {
    const size_t sz = (size_t)n * n;
    std::vector<double> h0(sz), hb(sz);
    srand(n);
    for (size_t i = 0; i < sz; ++i) h0[i] = (rand() % 2001 - 1000) * 1e-3;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) hb[(size_t)i * n + j] = (i == j ? 0.9 : 0.0) + (rand() % 2001 - 1000) * 1e-6;
    double *b0, *b1, *B;
    int *ctr, *tmo;
    CK(hipMalloc(&b0, sz * 8));
    CK(hipMalloc(&b1, sz * 8));
    CK(hipMalloc(&B, sz * 8));
    CK(hipMalloc(&ctr, (steps + 1) * sizeof(int)));
    CK(hipMalloc(&tmo, sizeof(int)));
    CK(hipMemcpy(B, hb.data(), sz * 8, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A = b0;
    g.B = B;
    g.C = b1;
    g.sA = g.sB = g.sC = 0;
    g.lda = g.ldb = g.ldc = n;
    g.M = g.N = g.K = n;
    g.alpha = 1.0;
    g.beta = 0;
    g.klo = KLO_ZERO;
    g.khi = KHI_FULL;
    g.lower_only = 0;
    g.tiles_m = g.tiles_n = n / 64;
    g.flags = 0;
    g.ntiles = (n / 64) * (n / 64);
    g.batch = 1;
    const int tiles = g.ntiles;
    std::vector<std::vector<double>> res(3, std::vector<double>(sz));
    double us[3] = {0, 0, 0};
    for (int variant = 0; variant < 3; ++variant) {
      for (int rep = 0; rep < 3; ++rep) {  // the last repetition is the one that counts (warm code, warm clocks)
        CK(hipMemcpy(b0, h0.data(), sz * 8, hipMemcpyHostToDevice));
        CK(hipMemset(ctr, 0, (steps + 1) * sizeof(int)));
        CK(hipMemset(tmo, 0, sizeof(int)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st));
        if (variant == 0) {
          for (int s = 0; s < steps; ++s) {
            GemmArgs gs = g;
            gs.A = (s & 1) ? b1 : b0;
            gs.C = (s & 1) ? b0 : b1;
            hipLaunchKernelGGL((gemm_kernel<double, false, true, 64, 4>), dim3(tiles, 1), dim3(256), 0, st, gs);
          }
        } else if (variant == 1) {
          hipLaunchKernelGGL((chain_kernel<1>), dim3(tiles), dim3(256), 0, st, g, b0, b1, steps, ctr, tmo);
        } else {
          hipLaunchKernelGGL((chain_kernel<2>), dim3(tiles), dim3(256), 0, st, g, b0, b1, steps, ctr, tmo);
        }
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[variant] = ms * 1e3 / steps;
        int t = 0;
        CK(hipMemcpy(&t, tmo, sizeof(int), hipMemcpyDeviceToHost));
        if (t) {
          printf("n=%d variant %d: a wait timed out\n", n, variant);
          return 1;
        }
      }
      CK(hipMemcpy(res[variant].data(), (steps & 1) ? b1 : b0, sz * 8, hipMemcpyDeviceToHost));
    }
    const bool eqB = res[0] == res[1], eqC = res[0] == res[2];
    printf("%6d %6d | %12.2f %12.2f %12.2f | B %s, C %s\n", n, tiles, us[0], us[1], us[2], eqB ? "==" : "DIFFERS",
           eqC ? "==" : "DIFFERS");
    CK(hipFree(b0));
    CK(hipFree(b1));
    CK(hipFree(B));
    CK(hipFree(ctr));
    CK(hipFree(tmo));
  }
  return 0;
}

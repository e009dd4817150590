// leaf_probe.hip -- where the time of one 128 x 128 leaf goes: per-panel shader-clock stamps of the diag wave and
// the update waves (leaf5), plus launch-to-launch durations of leaf3 / leaf5 for 1 and 16 blocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGPC_LEAF_TRACE] -I gpyreg_amd/csrc -o tools/leaf_probe tools/leaf_probe.hip
// (without the macro: durations only, of the kernel exactly as the library runs it)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "leaf.h"
using namespace gpc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifdef GPC_LEAF_TRACE
__global__ void set_trace(long long* p) { g_leaf_trace = p; }
#endif
__global__ void empty_kernel() {}
int main() {
  const int n = TILE, B = 16;
  std::vector<double> A((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) * 100.0 + (i == j ? 1.0 : 0.0);
  double *dA, *dA0, *dW, *dlog;
  int* dinfo;
  long long* dtr;
  CK(hipMalloc(&dA, sizeof(double) * n * n * B));
  CK(hipMalloc(&dA0, sizeof(double) * n * n * B));
  CK(hipMalloc(&dW, sizeof(double) * n * n * B));
  CK(hipMalloc(&dlog, sizeof(double) * B));
  CK(hipMalloc(&dinfo, sizeof(int) * B));
  CK(hipMalloc(&dtr, sizeof(long long) * 4 * 8 * 8));
  for (int b = 0; b < B; ++b) CK(hipMemcpy(dA0 + (size_t)b * n * n, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
  CK(hipMemset(dlog, 0, sizeof(double) * B));
  CK(hipMemset(dinfo, 0, sizeof(int) * B));
  CK(hipMemset(dtr, 0, sizeof(long long) * 256));
#ifdef GPC_LEAF_TRACE
  hipLaunchKernelGGL(set_trace, dim3(1), dim3(1), 0, 0, dtr);
#endif
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(256), 0, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("empty kernel: %.1f us (event to event: subtract from the figures below)\n", best * 1e3);
  }
  for (int ver : {3, 5})
    for (int blocks : {1, 16}) {
      float best = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemcpy(dA, dA0, sizeof(double) * n * n * B, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        if (ver == 3)
          hipLaunchKernelGGL((leaf3_kernel<double, false>), dim3(blocks), dim3(256), 0, 0, dA, (long long)n * n, n, dW, (long long)n * n, n, 0, dlog, dinfo, n);
        else
          hipLaunchKernelGGL((leaf5_kernel<double>), dim3(blocks), dim3(256), 0, 0, dA, (long long)n * n, n, dW, (long long)n * n, n, 0, dlog, dinfo, n, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      printf("leaf%d blocks=%2d: %.1f us (event to event, best of 20)\n", ver, blocks, best * 1e3);
    }
  int info[B];
  CK(hipMemcpy(info, dinfo, sizeof(info), hipMemcpyDeviceToHost));
  printf("info[0]=%d\n", info[0]);
#ifndef GPC_LEAF_TRACE
  return 0;
#endif
  long long tr[4][8][8];
  CK(hipMemcpy(tr, dtr, sizeof(tr), hipMemcpyDeviceToHost));
  const long long t0 = tr[0][0][0];
  printf("leaf5 stamps (shader-clock cycles from the diag wave's first stamp)\n");
  printf("diag wave : P  waitA_begin  chain_begin  chain_end  posted+L stored  after barrier\n");
  for (int P = 0; P < 8; ++P) {
    printf("  P=%d", P);
    for (int k = 0; k < 5; ++k) printf(" %8lld", tr[0][P][k] - t0);
    printf("   chain %lld  waitA %lld\n", tr[0][P][2] - tr[0][P][1], tr[0][P][1] - tr[0][P][0]);
  }
  for (int w = 1; w < 4; ++w) {
    printf("update wave %d: P | waitW  fast  solve+Wrow  waitL  Aupd+waitW  Vupd(to next panel)\n", w - 1);
    for (int P = 0; P < 8; ++P) {
      const long long* t = tr[w][P];
      const long long nxt = P < 7 ? tr[w][P + 1][0] : t[5];
      if (t[4] == 0) { printf("  P=%d begin %8lld | %6lld %6lld %6lld (no rows below)\n", P, t[0] - t0, t[1] - t[0], t[2] - t[1], t[3] - t[2]); continue; }
      printf("  P=%d begin %8lld | %6lld %6lld %6lld %6lld %6lld %6lld\n", P, t[0] - t0, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3],
             t[5] - t[4], nxt - t[5]);
    }
  }
  return 0;
}

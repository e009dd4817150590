// relative error of v_rsq_f64 and of 1 / 2 Newton steps on it (how many steps does the leaf's pivot need?)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = x[i];
  double r0 = __builtin_amdgcn_rsq(s);
  double r1 = r0 * (1.5 - 0.5 * s * r0 * r0);
  double r2 = r1 * (1.5 - 0.5 * s * r1 * r1);
  // fma form of one step: r1f = r0 + r0 * (0.5 - 0.5 s r0^2) computed with fma residual
  double e = fma(-s * r0, r0, 1.0);
  double r1f = fma(r0 * 0.5, e, r0);
  o[4 * i] = r0; o[4 * i + 1] = r1; o[4 * i + 2] = r2; o[4 * i + 3] = r1f;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(4 * n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = std::exp(((st >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 40.0); }
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
  double m[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) { long double ref = 1.0L / sqrtl((long double)x[i]); for (int j = 0; j < 4; ++j) { double e = (double)fabsl((o[4 * i + j] - ref) / ref); if (e > m[j]) m[j] = e; } }
  printf("max relative error: v_rsq_f64 %.3e | one Newton step %.3e | two %.3e | one step, fma residual form %.3e\n", m[0], m[1], m[2], m[3]);
  return 0;
}

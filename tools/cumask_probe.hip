// which (XCC, SE, CU) does each bit of a hipExtStreamCreateWithCUMask mask enable?  (diagnostic)
// Every mask tried here keeps >= 1 CU in every XCC under both plausible bit orders
// (interleaved: xcc = bit % 8; blocked: xcc = bit / 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // a little work so blocks spread over the enabled CUs
  unsigned long long t0 = clock64();
  while (clock64() - t0 < 20000) {}
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int run(const char* tag, const uint32_t* mask) {
  hipStream_t st;
  if (mask) CK(hipExtStreamCreateWithCUMask(&st, 8, mask)); else CK(hipStreamCreate(&st));
  const int nb = 4096;
  unsigned* d; CK(hipMalloc(&d, nb * 8));
  hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, st, d);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned> h(2 * nb);
  CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
  std::set<unsigned> per_xcc[16];
  for (int i = 0; i < nb; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%s:", tag);
  int tot = 0;
  for (int x = 0; x < 8; ++x) { printf(" xcc%d=%zu", x, per_xcc[x].size()); tot += per_xcc[x].size(); }
  printf("  total %d\n   xcc0 (se.sh.cu):", tot);
  for (unsigned v : per_xcc[0]) printf(" %u.%u.%u", v >> 8, (v >> 4) & 1, v & 0xf);
  printf("\n   xcc1 (se.sh.cu):");
  for (unsigned v : per_xcc[1]) printf(" %u.%u.%u", v >> 8, (v >> 4) & 1, v & 0xf);
  printf("\n");
  CK(hipFree(d));
  CK(hipStreamDestroy(st));
  return 0;
}
int main() {
  if (run("no mask", nullptr)) return 1;
  uint32_t m[8];
  // bits [32k, 32k+8): 64 CUs
  for (int k = 0; k < 8; ++k) m[k] = 0xffu;
  if (run("bits 32k..32k+7", m)) return 1;
  // complement
  for (int k = 0; k < 8; ++k) m[k] = ~0xffu;
  if (run("complement", m)) return 1;
  // discriminates the bit order: blocked -> xcc0 gets 32 CUs, interleaved -> 11 CUs in every XCC
  for (int k = 0; k < 8; ++k) m[k] = 0xffu;
  m[0] = 0xffffffffu;
  if (run("word0 full, others 0xff", m)) return 1;
  return 0;
}

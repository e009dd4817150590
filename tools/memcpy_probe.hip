#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
  hipStream_t st; hipStreamCreate(&st);
  size_t sizes[] = {16<<10, 64<<10, 256<<10, 1<<20, 4<<20, 8<<20, 64<<20, 128<<20};
  void* d; hipMalloc(&d, 128<<20);
  void* pin; hipHostMalloc(&pin, 128<<20, hipHostMallocDefault);
  for (size_t n : sizes) {
    std::vector<char> h(n, 1);
    for (int dir = 0; dir < 2; ++dir) for (int pinned = 0; pinned < 2; ++pinned) {
      void* hp = pinned ? pin : (void*)h.data();
      double best = 1e9, first = 0;
      for (int it = 0; it < 5; ++it) {
        std::vector<char> fresh;
        if (!pinned) { fresh.assign(n, 2); hp = fresh.data(); }  // a new allocation every time, as a caller's arrays are
        auto t0 = std::chrono::steady_clock::now();
        if (dir == 0) hipMemcpyAsync(d, hp, n, hipMemcpyHostToDevice, st);
        else hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best) best = ms;
        if (it == 0) first = ms;
      }
      printf("%8zu KB %s %s: best %8.3f ms (%7.2f GB/s) first %8.3f ms\n", n >> 10, dir ? "D2H" : "H2D", pinned ? "pinned      " : "pageable-new", best, n / best / 1e6, first);
    }
    // host memcpy rate
    std::vector<char> h2(n);
    auto t0 = std::chrono::steady_clock::now();
    memcpy(h2.data(), h.data(), n);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%8zu KB host memcpy: %8.3f ms %7.2f GB/s\n", n >> 10, ms, n / ms / 1e6);
  }
  return 0;
}

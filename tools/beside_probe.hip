// beside_probe: what does a low-register wave get while the W^T W launch (gemm.h: gemm_persist_kernel<double, true,
// true, 128, 4>, two resident blocks per CU, 2 x 240 of 512 VGPRs per lane) has the chip?  The two triangular mat-vecs
// of a gradient evaluation run there (gpcore.hip: solves beside lauum) and take 20x their stand-alone time; this
// measures WHY, with 30-VGPR probe kernels of the mat-vec's launch shape (4096 blocks of 256 threads) run alone and
// beside the real launch (N = 4096, 16 samples):
//   mode 0  one 16-byte buffer load per lane at a time (fresh lines), waited for: the load round trip
//   mode 1  four such loads in flight, then waited for (the mat-vec's pattern)
//   mode 2  64 dependent v_fma_f64 per iteration: fp64 VALU issue beside the MFMAs
//   mode 3  64 dependent v_add_u32 per iteration: integer VALU issue
//   mode 4  64 dependent s_add_u32 per iteration: scalar issue (does the wave get issue slots at all?)
//   mode 5  the mat-vec's own loop (blas1.h: trmv_low_kernel): four rows of W per load of r, 8 FMAs, 4 KB per iteration
//   mode 6  the same bytes as one row per wave (four iterations of 1 KB): fits 16 VGPRs, two such waves per SIMD
// Per mode: shader cycles per iteration (s_memtime), the life time of a block (s_memrealtime, 100 MHz), the number of
// blocks resident at once (sum of life times / span of the launch) and the launch's duration.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/beside_probe.hip -o beside_probe
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../gpyreg_amd/csrc/gemm.h"

using namespace gpc;

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

struct Rec {
  unsigned long long t0, t1;  // s_memrealtime at the start / end of the block's first wave
  unsigned long long cyc;     // s_memtime cycles of the loop
  unsigned hwid, pad;
};

template <int MODE, int CLAIM>
__global__ __launch_bounds__(256) void probe_kernel(const double* __restrict__ src, size_t nelem, int iters, int prio,
                                                    Rec* rec, double* sink) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  if (CLAIM) asm volatile("" ::: "v29");  // 32 VGPRs as allocated, like the mat-vec: one such wave per SIMD beside the GEMM
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t wave = (size_t)blockIdx.x * 4 + wv;
  // every wave streams its own 16-byte-per-lane lines (1 KB per load instruction), never re-read
  const size_t per_wave = (size_t)iters * 4 * 128;  // doubles
  const double* base = src + (wave * per_wave) % (nelem - per_wave);
  const __amdgpu_buffer_rsrc_t q = make_rsrc(base);
  double acc = 0.0;
  unsigned iacc = lane;
  unsigned sacc = 1;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(q, (unsigned)(it * 4096 + lane * 16), 0, 0);
      acc += reinterpret_cast<const double*>(&a)[0];
    }
  } else if (MODE == 1) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      const unsigned o = (unsigned)(it * 4096 + lane * 16);
      const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(q, o, 0, 0);
      const u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(q, o + 1024, 0, 0);
      const u32x4 a2 = __builtin_amdgcn_raw_buffer_load_b128(q, o + 2048, 0, 0);
      const u32x4 a3 = __builtin_amdgcn_raw_buffer_load_b128(q, o + 3072, 0, 0);
      acc += reinterpret_cast<const double*>(&a0)[0] + reinterpret_cast<const double*>(&a1)[1] +
             reinterpret_cast<const double*>(&a2)[0] + reinterpret_cast<const double*>(&a3)[1];
    }
  } else if (MODE == 2) {
    double x = 1.0 + lane * 1e-9;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(acc) : "v"(x));
    }
  } else if (MODE == 3) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) asm volatile("v_add_u32 %0, %0, %0" : "+v"(iacc));
    }
  } else if (MODE == 4) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) asm volatile("s_add_u32 %0, %0, %0" : "+s"(sacc));
    }
  } else if (MODE == 5) {  // blas1.h: trmv_low_kernel's loop: four rows of W (16 bytes per lane each) per load of r
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      const unsigned o = (unsigned)(it * 4096 + lane * 16);
      const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(q, o, 0, 0);
      const u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(q, o + 1024, 0, 0);
      const u32x4 a2 = __builtin_amdgcn_raw_buffer_load_b128(q, o + 2048, 0, 0);
      const u32x4 a3 = __builtin_amdgcn_raw_buffer_load_b128(q, o + 3072, 0, 0);
      const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(q, (unsigned)(lane * 16), 0, 0);
      const double r0 = reinterpret_cast<const double*>(&rr)[0], r1 = reinterpret_cast<const double*>(&rr)[1];
      acc += reinterpret_cast<const double*>(&a0)[0] * r0;
      s1 += reinterpret_cast<const double*>(&a1)[0] * r0;
      s2 += reinterpret_cast<const double*>(&a2)[0] * r0;
      s3 += reinterpret_cast<const double*>(&a3)[0] * r0;
      acc += reinterpret_cast<const double*>(&a0)[1] * r1;
      s1 += reinterpret_cast<const double*>(&a1)[1] * r1;
      s2 += reinterpret_cast<const double*>(&a2)[1] * r1;
      s3 += reinterpret_cast<const double*>(&a3)[1] * r1;
    }
    acc += s1 + s2 + s3;
  } else {  // MODE 6: one row per wave, the same bytes per wave in four iterations: fits 16 VGPRs (two waves per SIMD)
#pragma unroll 1
    for (int it = 0; it < 4 * iters; ++it) {
      const unsigned o = (unsigned)(it * 1024 + lane * 16);
      const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(q, o, 0, 0);
      const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(q, (unsigned)(lane * 16), 0, 0);
      acc += reinterpret_cast<const double*>(&a0)[0] * reinterpret_cast<const double*>(&rr)[0];
      acc += reinterpret_cast<const double*>(&a0)[1] * reinterpret_cast<const double*>(&rr)[1];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (acc == 12345.678 || iacc == 0x7fffffffu || sacc == 0x7ffffffu) sink[0] = acc + iacc + sacc;  // keep the chains
  if (threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    Rec r;
    r.t0 = t0;
    r.t1 = __builtin_amdgcn_s_memrealtime();
    r.cyc = c1 - c0;
    r.hwid = hw;
    r.pad = 0;
    rec[blockIdx.x] = r;
  }
}

template <int MODE, int CLAIM>
static void run_mode(hipStream_t sp, hipStream_t sg, const GemmArgs& g, int* ctr, const double* src, size_t nelem,
                     int blocks, int iters, int prio, Rec* drec, double* sink, bool beside, const char* what) {
  hipEvent_t e0, e1, g0, g1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&g0));
  CK(hipEventCreate(&g1));
  CK(hipMemset(ctr, 0, CTR_STRIDE * sizeof(int)));
  CK(hipDeviceSynchronize());
  if (beside) {
    CK(hipEventRecord(g0, sg));
    CK(launch_gemm<double>(sg, g, true, true, 16, 0, ctr, nullptr));
    CK(hipEventRecord(g1, sg));
    usleep(600);  // the probe meets the GEMM in its steady state, not while its first operands are on their way
  }
  CK(hipEventRecord(e0, sp));
  hipLaunchKernelGGL((probe_kernel<MODE, CLAIM>), dim3(blocks), dim3(256), 0, sp, src, nelem, iters, prio, drec, sink);
  CK(hipGetLastError());
  CK(hipEventRecord(e1, sp));
  CK(hipDeviceSynchronize());
  float ms = 0, gms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  if (beside) CK(hipEventElapsedTime(&gms, g0, g1));
  std::vector<Rec> rec(blocks);
  CK(hipMemcpy(rec.data(), drec, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  double life = 0, cyc = 0;
  for (const Rec& r : rec) {
    tmin = std::min(tmin, r.t0);
    tmax = std::max(tmax, r.t1);
    life += (double)(r.t1 - r.t0);
    cyc += (double)r.cyc;
  }
  const double span_us = (tmax - tmin) / 100.0, life_us = life / blocks / 100.0;
  printf("%-30s %2d VGPRs %-7s prio %d | launch %8.3f ms | %9.0f cycles/iteration | block life %8.2f us | resident %6.1f blocks",
         what, CLAIM ? 32 : 16, beside ? "BESIDE" : "alone", prio, ms, cyc / blocks / iters, life_us, life / 100.0 / span_us);
  if (beside) printf(" | W^T W %6.3f ms", gms);
  printf("\n");
}

int main(int argc, char** argv) {
  const int npad = 4096, batch = 16;
  const size_t sM = (size_t)npad * npad;
  double *W, *out, *sink;
  int* ctr;
  Rec* drec;
  CK(hipMalloc(&W, sM * batch * 8));
  CK(hipMalloc(&out, sM * batch * 8));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&ctr, CTR_STRIDE * sizeof(int)));
  const int blocks = 4096;
  CK(hipMalloc(&drec, blocks * sizeof(Rec)));
  {
    std::vector<double> h(sM);
    srand(1);
    for (int i = 0; i < npad; ++i)
      for (int j = 0; j < npad; ++j) h[(size_t)i * npad + j] = j <= i ? (rand() % 2001 - 1000) * 1e-4 : 0.0;
    for (int b = 0; b < batch; ++b) CK(hipMemcpy(W + b * sM, h.data(), sM * 8, hipMemcpyHostToDevice));
  }
  // a separate 2 GiB source for the probe's streams (the mat-vecs read W itself; the lines are not shared either way)
  const size_t nelem = (size_t)1 << 28;
  double* src;
  CK(hipMalloc(&src, nelem * 8));
  CK(hipMemset(src, 0, nelem * 8));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_block_slots = 2 * prop.multiProcessorCount;
  GemmArgs g;
  g.A = W;
  g.B = W;
  g.C = out;
  g.sA = g.sB = g.sC = (long long)sM;
  g.lda = g.ldb = g.ldc = npad;
  g.M = g.N = g.K = npad;
  g.alpha = 1.0;
  g.beta = 0;
  g.klo = KLO_ROW;
  g.khi = KHI_FULL;
  g.lower_only = 1;
  g.tiles_n = npad / TILE;
  hipStream_t sp, sg;
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
  // warm-up of the GEMM (code, clocks)
  for (int i = 0; i < 2; ++i) {
    CK(hipMemset(ctr, 0, CTR_STRIDE * sizeof(int)));
    CK(launch_gemm<double>(sg, g, true, true, batch, 0, ctr, nullptr));
    CK(hipDeviceSynchronize());
  }
  const int it_ld = argc > 1 ? atoi(argv[1]) : 16;
  for (int i = 0; i < 3; ++i) {  // clocks: the first GEMMs of a process run 10-20 % slower
    CK(hipMemset(ctr, 0, CTR_STRIDE * sizeof(int)));
    CK(launch_gemm<double>(sg, g, true, true, batch, 0, ctr, nullptr));
    CK(hipDeviceSynchronize());
  }
  for (int beside = 0; beside < 2; ++beside)
    for (int prio = 0; prio < 2; ++prio) {
      if (!beside && prio) continue;
      run_mode<0, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "0 one load in flight");
      run_mode<1, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "1 four loads in flight");
      run_mode<2, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "2 64 dependent v_fma_f64");
      run_mode<3, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "3 64 dependent v_add_u32");
      run_mode<4, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "4 64 dependent s_add_u32");
      run_mode<5, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "5 mat-vec loop, 4 rows");
      run_mode<6, 0>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "6 mat-vec loop, 1 row");
      run_mode<6, 1>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "6 mat-vec loop, 1 row");
      // the same with the registers the probe really needs (8-16: two to four such waves per SIMD beside the GEMM)
      run_mode<0, 0>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "0 one load in flight");
      run_mode<1, 0>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "1 four loads in flight");
      run_mode<2, 0>(sp, sg, g, ctr, src, nelem, blocks, it_ld, prio, drec, sink, beside, "2 64 dependent v_fma_f64");
    }
  return 0;
}

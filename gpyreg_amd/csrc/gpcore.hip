// gpcore.hip -- libgpcore.so: C ABI (include/gpcore.h) over the HIP kernels.
//
// Path implemented (reference lines in parentheses):
//   covariance.compute              (covariance_functions.py:135-367, isotropic_...:104-221)
//   GP.__core_computation           (gaussian_process.py:2357-2521)
//   GP.update full-recompute loop   (gaussian_process.py:870-884)   -> gpc_posterior_batch
//   GP.predict K* solves            (gaussian_process.py:1741-1764) -> gpc_predict
// Data layout in HBM: every N x N matrix is row-major, padded to a multiple of 128 with
// an identity block (exactly inert for Cholesky, inverse, log-det and traces), one slab
// per hyperparameter sample, samples contiguous (batch stride = npad^2).
#include "../../include/gpcore.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "blas1.h"
#include "common.h"
#include "covfun.h"
#include "gemm.h"
#include "leaf.h"
#include "plan.h"
#ifdef GPC_EXPERIMENTS
// Schedules that were built, measured and rejected -- the tile-level dataflow graph, independent
// per-sample pipelines, rectangular / eight-wave tiles, right-looking panels -- are compiled only into the experiments
// build (gpyreg_amd/build.py --experiments -> lib/libgpcore_exp.so), which tests/ and tools/ opt into through
// GPYREG_AMD_LIB.  The product library contains the shipped stream-ordered schedule and nothing else.
#include "dag.h"
#endif

using namespace gpc;

namespace {

std::string g_create_err;

// bumped whenever a device buffer is (re)allocated: cached launch graphs hold raw pointers
inline unsigned long long g_alloc_epoch = 0;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t need) {
    if (need <= bytes) return hipSuccess;
    ++g_alloc_epoch;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e == hipSuccess) bytes = need;
    return e;
  }
  // for buffers no captured launch graph refers to (a posterior's private constants): no epoch bump
  hipError_t ensure_private(size_t need) {
    if (need <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e == hipSuccess) bytes = need;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename U>
  U* as() const {
    return reinterpret_cast<U*>(p);
  }
};

// Page-locked host staging for transfers between the caller's (pageable) arrays and HBM.
// Copies straight from/to pageable memory go through the runtime's pin-on-demand path, measured
// at ~20 ms of fixed host time for the first multi-MB copy of a call (S=1024, N=500: 30 ms per
// batched evaluation, 6 ms with staging); staging through long-lived pinned blocks costs a host
// memcpy (~30 GB/s) and makes the copy truly asynchronous.  Lifetime: begin() at the top of an
// entry point (the stream is idle there: every entry point ends with a sync), finish() after
// the final sync delivers the downloads.
// Gathered transfers of one call.  The boundary hands over 5-7 small host arrays per call (hyperparameter rows, mean,
// noise, their gradients' inputs) and takes back 3-6 (log det, quadratic form, info, gradients): as separate
// hipMemcpyAsync's each is a blit kernel of its own on the stream (~10 us apiece, 100-190 us per call against
// 250 us of device work for a single evaluation at N = 200).  Here the host side stages everything in ONE pinned
// block and one kernel moves the segments (the pinned block is mapped: the kernel reads / writes host memory over
// the bus directly) and zeroes the scalar block on the way.
struct XferSeg {
  void* dst;
  const void* src;
  unsigned long long n8;  // 8-byte words
};
struct XferDesc {
  static constexpr int MAXSEG = 8;
  XferSeg seg[MAXSEG];
  int nseg;
  void* zero;
  unsigned long long zero8;
  // optional: the scaled inputs of all samples (covfun.h: scale_x_kernel's arithmetic), mul / dv read from the
  // staged host copies, so that the first kernel of the pipeline is the covariance build itself
  const double* X = nullptr;
  const double* mul = nullptr;
  const double* dv = nullptr;
  double* xs = nullptr;
  int n = 0, npad = 0, D = 0, cnt = 0;
  // optional (gpc_nll_batch_cm: constant mean, and / or scalar noise): what the host would otherwise compute per point
  // and upload -- r[b][i] = y[i] - m0[b] from the RESIDENT y and the per-sample constant, dvec[b][i] = dval[b] (1 in
  // the identity padding) -- formed here; m0 / dval are staged host copies (one value per sample)
  const double* y = nullptr;
  const double* m0 = nullptr;
  double* r_out = nullptr;
  const double* dval = nullptr;
  double* dvec_out = nullptr;
  int fn = 0, fnpad = 0, fcnt = 0;
  // optional (the download launch of a call, round 6): the launch announces its own completion in host memory -- every
  // block makes its stores visible system-wide and counts itself in `done_ctr` (a device word the upload launch of the same
  // call zeroed), the last one stores `seq` into `flag` -- so that the host can watch that word instead of waiting for the
  // runtime's end-of-kernel signal (several microseconds later)
  unsigned long long* flag = nullptr;
  unsigned long long seq = 0;
  int* done_ctr = nullptr;
};
__device__ __forceinline__ void xfer_body_noxs(const XferDesc& d, unsigned long long gt, unsigned long long stride) {
  // the segments as ONE index space: a thread's words are independent loads from (mapped) host memory, all in flight
  // together -- a loop per segment made nseg dependent PCIe round trips of ~2.5 us each out of a small upload
  unsigned long long total = 0;
  for (int k = 0; k < d.nseg; ++k) total += d.seg[k].n8;
  // ... dealt to the blocks in contiguous chunks of equal size (a small upload read by the first block alone queues
  // behind that one CU's outstanding host reads: 8 us for 2.6 KB against 3.6 us for a third of it)
  const unsigned long long nblk = stride / 256, blk = gt / 256, lt = gt % 256;
  const unsigned long long chunk = (total + nblk - 1) / nblk;
  const unsigned long long hi = chunk * (blk + 1) < total ? chunk * (blk + 1) : total;
  for (unsigned long long i = chunk * blk + lt; i < hi; i += 256) {
    unsigned long long j = i;
    int k = 0;
    while (j >= d.seg[k].n8) {
      j -= d.seg[k].n8;
      ++k;
    }
    reinterpret_cast<unsigned long long*>(d.seg[k].dst)[j] = reinterpret_cast<const unsigned long long*>(d.seg[k].src)[j];
  }
  unsigned long long* z = reinterpret_cast<unsigned long long*>(d.zero);
  for (unsigned long long i = gt; i < d.zero8; i += stride) z[i] = 0ull;
  if (d.r_out || d.dvec_out) {
    const unsigned long long tot = (unsigned long long)d.fnpad * d.fcnt;
    for (unsigned long long q = gt; q < tot; q += stride) {
      // the sample index is the same for the 64 lanes of a wave (gt is wave-contiguous from a multiple of 64, stride
      // a multiple of 256, fnpad a multiple of 128): the two per-sample constants are read from the mapped host block
      // ONCE per wave through the scalar unit, not once per element (S * npad uncached PCIe reads)
      const int b = __builtin_amdgcn_readfirstlane((int)(q / d.fnpad));
      const int i = (int)(q - (unsigned long long)b * d.fnpad);
      if (d.r_out) {
        const double m0b = d.m0[b];
        d.r_out[q] = i < d.fn ? d.y[i] - m0b : 0.0;
      }
      if (d.dvec_out) {
        const double dvb = d.dval[b];
        d.dvec_out[q] = i < d.fn ? dvb : 1.0;
      }
    }
  }
}
__device__ __forceinline__ void xfer_body(const XferDesc& d, unsigned long long gt, unsigned long long stride) {
  xfer_body_noxs(d, gt, stride);
  if (d.xs) {
    const unsigned long long per = (unsigned long long)d.npad * d.D, tot = per * d.cnt;
    for (unsigned long long q = gt; q < tot; q += stride) {
      const unsigned long long b = q / per, idx = q - b * per;
      const int i = (int)(idx / d.D), h = (int)(idx % d.D);
      double v = 0.0;
      if (i < d.n) v = d.X[(size_t)i * d.D + h] * d.mul[b * d.D + h] / d.dv[b * d.D + h];
      d.xs[q] = v;
    }
  }
}
__global__ __launch_bounds__(256) void xfer_kernel(XferDesc d) {
  xfer_body(d, (unsigned long long)blockIdx.x * 256 + threadIdx.x, (unsigned long long)gridDim.x * 256);
  if (d.flag) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      // (done_ctr == nullptr: a launch of ONE workgroup -- it is its own last one)
      const int before = d.done_ctr ? __hip_atomic_fetch_add(d.done_ctr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) : 0;
      if (before == (int)gridDim.x - 1) __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Problems that are one 128 x 128 leaf (N <= 128): the upload kernel and the covariance build in ONE launch.
// grid = (3 lower 64-tiles, samples).  Every block takes its share of the gathered upload (xfer_body: the segments land
// in their device buffers for the kernels that follow), then scales the 128 rows of its sample's inputs by itself --
// every block of a sample writes the same values: each needs rows of both halves -- and builds its tile, reading the
// per-sample scalars and the diagonal term from the STAGED HOST COPIES (`sp_h`, `dvec_h`: mapped pinned memory; the
// device copies are being written by other blocks of this very launch and are not ordered against this read).
template <typename T, int KIND, int DEG>
__global__ __launch_bounds__(256) void small_front_kernel(XferDesc d, CovDesc cd, const double* __restrict__ sp_h,
                                                          const double* __restrict__ dvec_h, int dvec_vec,
                                                          T* __restrict__ A_all, long long sA) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  // this sample's host-resident inputs, fetched ONCE: a load from mapped host memory is a PCIe round trip (~2.5 us),
  // and the code below would otherwise make four dependent ones (scaling factors, scalars, diagonal term)
  constexpr int DL = 64;  // input dimensions whose scaling factors are kept in LDS (more: read from the host copy)
  __shared__ double h_sp[SP_STRIDE], h_dvec[TILE], h_mul[DL], h_dv[DL];
  const int b = blockIdx.y, t = threadIdx.x;
#ifdef GPC_FRONT_TRACE
  long long ts[6];
  ts[0] = wall_clock64();
#endif
  const bool dl = d.D <= DL;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  // the diagonal term is read by the diagonal tiles only (tile 0: rows 0..63, tile 2: rows 64..127)
  const int drow = blockIdx.x == 2 ? CT : 0;
  const bool dg = blockIdx.x != 1 && t < CT;
  // (dvec_vec = 0: one value per sample -- scalar noise; the padding rows carry 1 either way)
  if (dg) v0 = dvec_vec ? dvec_h[(size_t)b * TILE + drow + t] : (drow + t < d.n ? dvec_h[b] : 1.0);
  if (t < SP_STRIDE) v1 = sp_h[(size_t)b * SP_STRIDE + t];
  if (dl && t < d.D) {
    v2 = d.mul[(size_t)b * d.D + t];
    v3 = d.dv[(size_t)b * d.D + t];
  }
  xfer_body_noxs(d, ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x,
                 (unsigned long long)gridDim.x * gridDim.y * 256);
#ifdef GPC_FRONT_TRACE
  ts[1] = wall_clock64();
#endif
  if (dg) h_dvec[drow + t] = v0;
  if (t < SP_STRIDE) h_sp[t] = v1;
  if (dl && t < d.D) {
    h_mul[t] = v2;
    h_dv[t] = v3;
  }
  __syncthreads();
#ifdef GPC_FRONT_TRACE
  ts[2] = wall_clock64();
#endif
  double* xs = d.xs + (size_t)b * d.npad * d.D;
  const double* mulp = dl ? h_mul : d.mul + (size_t)b * d.D;
  const double* dvp = dl ? h_dv : d.dv + (size_t)b * d.D;
  for (int q = t; q < d.npad * d.D; q += 256) {
    const int i = q / d.D, h = q - i * d.D;
    xs[q] = i < d.n ? d.X[(size_t)i * d.D + h] * mulp[h] / dvp[h] : 0.0;
  }
  __syncthreads();  // (a block reads back only what it wrote itself)
#ifdef GPC_FRONT_TRACE
  ts[3] = wall_clock64();
#endif
  // sample index 0 for the scalars and the diagonal term: the LDS copies hold this block's sample only
  build_tile_at<T, KIND, DEG>(cd, d.xs, h_sp, h_dvec, d.n, d.npad, A_all, sA, d.npad, blockIdx.x, b, 0, xi, xj);
#ifdef GPC_FRONT_TRACE
  __syncthreads();
  ts[4] = wall_clock64();
  if (t == 0 && b == 0)
    printf("front tile %d: copy %lld  lds %lld  xs %lld  build %lld  (10 ns ticks)\n", (int)blockIdx.x, ts[1] - ts[0],
           ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3]);
#endif
}

struct PinBuf {
  struct Blk {
    char* p;
    size_t bytes, used;
  };
  struct Pending {
    void* dst;
    const void* src;
    size_t n;
  };
  std::vector<Blk> blks;
  std::vector<Pending> pend;
  static constexpr size_t kMin = 32u << 10;    // smaller copies: the runtime's own staging is fine
  static constexpr size_t kMax = 512ull << 20; // larger ones amortise the pin-on-demand cost themselves

  void begin() {
    pend.clear();
    if (blks.size() > 1) {  // the previous call outgrew its block: one block of the total next time
      size_t tot = 0;
      for (Blk& b : blks) {
        tot += b.bytes;
        (void)hipHostFree(b.p);
      }
      blks.clear();
      (void)alloc(tot);
    }
    for (Blk& b : blks) b.used = 0;
  }
  void* alloc(size_t n) {
    n = (n + 255) & ~(size_t)255;
    if (!blks.empty() && blks.back().used + n <= blks.back().bytes) {
      void* r = blks.back().p + blks.back().used;
      blks.back().used += n;
      return r;
    }
    Blk b{nullptr, std::max(n + n / 2, (size_t)4 << 20), 0};
    if (hipHostMalloc(reinterpret_cast<void**>(&b.p), b.bytes, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    b.used = n;
    blks.push_back(b);
    return b.p;
  }
  hipError_t up(void* dst, const void* src, size_t n, hipStream_t st) {
    if (n >= kMin && n <= kMax)
      if (void* h = alloc(n)) {
        memcpy(h, src, n);
        src = h;
      }
    return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st);
  }
  // the data lands in dst at finish(), which the caller runs after synchronising the stream
  hipError_t down(void* dst, const void* src, size_t n, hipStream_t st) {
    if (n >= kMin && n <= kMax)
      if (void* h = alloc(n)) {
        pend.push_back({dst, h, n});
        dst = h;
      }
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st);
  }
  void finish() {
    for (Pending& q : pend) memcpy(q.dst, q.src, q.n);
    pend.clear();
  }
  // ---- gathered form: stage() any number of uploads (<= MAXSEG), flush_up() moves them with one kernel;
  // gather() downloads likewise, flush_down() + (after the stream is synchronised) finish().
  // Falls back to individual copies when a segment is not 8-byte sized, the total is large enough for the copy
  // engines to win (kGather), or pinned memory is not to be had.
  static constexpr size_t kGather = 2u << 20;
  XferDesc upd{}, downd{};
  std::vector<Pending> up_fallback;
  void begin_gather() {
    upd.nseg = downd.nseg = 0;
    upd.xs = nullptr;
    upd.r_out = upd.dvec_out = nullptr;
    upd.zero = downd.zero = nullptr;
    upd.zero8 = downd.zero8 = 0;
    up_fallback.clear();
    gathered_bytes = 0;
    down_plain = false;
  }
  size_t gathered_bytes = 0;
  // returns the staged (pinned, device-readable) copy, or nullptr when the segment travels by itself
  const void* stage(void* dst, const void* src, size_t n) {
    if (n == 0) return nullptr;
    void* h = (n % 8 == 0 && upd.nseg < XferDesc::MAXSEG && gathered_bytes + n <= kGather) ? alloc(n) : nullptr;
    if (!h) {
      up_fallback.push_back({dst, src, n});
      return nullptr;
    }
    memcpy(h, src, n);
    upd.seg[upd.nseg++] = {dst, h, (unsigned long long)(n / 8)};
    gathered_bytes += n;
    return h;
  }
  hipError_t flush_up(hipStream_t st, void* zero, size_t zero_bytes) {
    for (Pending& q : up_fallback) {
      hipError_t e = up(q.dst, q.src, q.n, st);
      if (e != hipSuccess) return e;
    }
    up_fallback.clear();
    if (zero_bytes % 8) {
      hipError_t e = hipMemsetAsync(zero, 0, zero_bytes, st);
      if (e != hipSuccess) return e;
      zero_bytes = 0;
    }
    upd.zero = zero;
    upd.zero8 = zero_bytes / 8;
    if (upd.nseg == 0 && upd.zero8 == 0 && !upd.xs && !upd.r_out && !upd.dvec_out) return hipSuccess;
    const unsigned long long words = gathered_bytes / 8 + upd.zero8 + (upd.xs ? (unsigned long long)upd.npad * upd.D * upd.cnt : 0ull) +
                                     ((upd.r_out || upd.dvec_out) ? (unsigned long long)upd.fnpad * upd.fcnt : 0ull);
    const int blocks = (int)std::min<unsigned long long>(256, (words + 255) / 256);
    hipLaunchKernelGGL(xfer_kernel, dim3(std::max(1, blocks)), dim3(256), 0, st, upd);
    upd.nseg = 0;
    return hipGetLastError();
  }
  // flush_up without the launch: the fallback copies go out, `out` receives the descriptor of the gathered segments
  // (the caller's own kernel moves them: small_front_kernel)
  hipError_t take_up(hipStream_t st, void* zero, size_t zero_bytes, XferDesc& out) {
    for (Pending& q : up_fallback) {
      hipError_t e = up(q.dst, q.src, q.n, st);
      if (e != hipSuccess) return e;
    }
    up_fallback.clear();
    if (zero_bytes % 8) {
      hipError_t e = hipMemsetAsync(zero, 0, zero_bytes, st);
      if (e != hipSuccess) return e;
      zero_bytes = 0;
    }
    upd.zero = zero;
    upd.zero8 = zero_bytes / 8;
    out = upd;
    upd.nseg = 0;
    return hipSuccess;
  }
  // dst: host; the data lands there at finish()
  bool down_plain = false;  // a download of this call travelled as a plain copy (the gathered launch does not cover it)
  hipError_t gather(void* dst, const void* src, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    void* h = (n % 8 == 0 && downd.nseg < XferDesc::MAXSEG && n <= kGather / 8) ? alloc(n) : nullptr;
    if (!h) {
      down_plain = true;
      return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st);
    }
    pend.push_back({dst, h, n});
    downd.seg[downd.nseg++] = {h, src, (unsigned long long)(n / 8)};
    return hipSuccess;
  }
  // flag / seq / done_ctr: see XferDesc (nullptr: the caller synchronises the stream)
  hipError_t flush_down(hipStream_t st, unsigned long long* flag = nullptr, unsigned long long seq = 0, int* done_ctr = nullptr) {
    if (downd.nseg == 0) return hipSuccess;
    unsigned long long words = 0;
    for (int k = 0; k < downd.nseg; ++k) words += downd.seg[k].n8;
    int blocks = (int)std::min<unsigned long long>(64, (words + 255) / 256);
    if (flag && !done_ctr) blocks = 1;  // (no zeroed counter at hand: one workgroup moves the few KB and announces itself)
    downd.flag = flag;
    downd.seq = seq;
    downd.done_ctr = done_ctr;
    hipLaunchKernelGGL(xfer_kernel, dim3(std::max(1, blocks)), dim3(256), 0, st, downd);
    downd.nseg = 0;
    downd.flag = nullptr;
    return hipGetLastError();
  }
  void release() {
    for (Blk& b : blks) (void)hipHostFree(b.p);
    blks.clear();
    pend.clear();
  }
};

}  // namespace

struct gpc_ctx {
  int device = 0;
  hipStream_t st = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  static constexpr int MAXG = 8;
  hipStream_t gst[MAXG] = {};  // sample-group streams
  hipStream_t sst[MAXG + 1] = {};  // side streams of the deferred inverse products (plan.h), one per group + main
  static constexpr int NDEV = 64;
  hipEvent_t dev_ev[MAXG + 1][NDEV] = {};
  hipEvent_t ev_bfork[MAXG + 1] = {}, ev_btail[MAXG + 1] = {};  // split kernel build (device_section)
  // GPC_DEFER_MIN: node size from which U = T21 W11 runs on the side stream (0 off, -1 auto: the top two
  // levels when the batch is small enough for latency-bound phases to matter); GPC_DEFER_RESERVE: CUs per
  // XCD the deferred launch keeps empty (2, 4, 8, 12)
  int defer_min = -1, defer_reserve = 8;
  // The device's own numbering of shader engines and CUs, probed at gpc_create (probe_cu_map): cu_seen[xcc * 8 + se]
  // = bit mask of the CU_IDs workgroups were observed on.  The table of reserved CUs (rsv_tbl, same indexing; read
  // by common.h: cu_reserve_bail) is derived from it, so nothing about the harvesting map is assumed.  cu_map_ok:
  // the probe saw every CU the runtime reports, the same number in every engine; otherwise the deferred schedule
  // (which needs empty CUs to pay) is off.
  unsigned cu_seen[64] = {};
  bool cu_map_ok = false;
  DevBuf rsv_tbl;
  // NLL-only evaluations: largest diagonal block that gets its inverse (plan.h: potrf_nll).  0: round-2 scheme (every
  // left child inverted); -1: automatic -- 512 from npad = 2048 on, else 0.  The choice depends on the problem size
  // ONLY, never on the batch: a row of a batch must carry the bits of its single evaluation (the speculative slice
  // sampler relies on it, slice_sample.py), and the two schemes differ in rounding.
  // Measured (round-3 sweep, script since removed; ms per batch at block 0 / 256 / 512 / 1024): cfg3 S=16 8.85 / 8.50 / 8.41 / 8.46;
  // N=8192 S=8 29.8 / 28.1 / 27.5 / 27.5; N=2048 S=16 1.98 / 1.95 / 1.89 / 2.00; single samples and small batches are
  // launch-bound and lose with the extra launches of the blocked solves (N=4096 S=1 2.28 / 2.42 / 2.30 / 2.28;
  // N=2048 S=1 0.96 / 1.01 / 0.98 / 0.98; N=1000 S=8 0.52 / 0.53 / 0.54 / 0.54): below npad = 2048 the round-2 scheme stays.
  int nll_block = -1;
  int solves_beside_lauum = 1;  // option: the two triangular mat-vecs of a gradient evaluation run under the W^T W launch
#ifdef GPC_EXPERIMENTS
  // right-looking panels with look-ahead for NLL-only evaluations (plan.h: potrf_rl): panel height; 0 = off (default)
  int rl_panel = 0;
  int rl_ahead_max = 8;  // look-ahead (side stream + reserved CUs) only for batches with S (npad/4096)^3 <= this
#endif
  int stable = 0;        // option: every factorization in stable mode (plan.h), not only the jitter retries
  int small_path = 1;    // option: problems of one 128 x 128 leaf take the two-launch pipeline (Pipe::small_section)
  // One-leaf evaluations without gradient (round 6): the host watches a word of a COHERENT pinned block that the last
  // block of the call writes, instead of hipStreamSynchronize (the runtime's completion path costs several microseconds
  // after the kernel's last store), and the four timing events of a call -- each a barrier packet on the stream -- are
  // recorded only when "small_timing" is set (gpc_last_timing then reports the device section; 0 otherwise).
  int small_poll = 1, small_timing = 0;
  double* land_blk = nullptr;            // hipHostMallocCoherent, LAND_BYTES
  static constexpr size_t LAND_BYTES = 64u << 10;
  unsigned long long land_seq = 0;
  unsigned long long small_polled = 0, small_synced = 0;  // statistics ("small_polled" / "small_synced")
  int check_queues = 0;  // debug option: verify after every pipeline that the tile queues of its persistent launches were drained
  hipEvent_t ev_up = nullptr, ev_done[MAXG] = {};
  int groups = 2;
  hipEvent_t ev_l0[MAXG + 1] = {}, ev_l1[MAXG + 1] = {};  // around the lauum launch of each group
  double ms_lauum = 0, flops_lauum = 0;  // slowest group's lauum launch of the last call
  int lauum_groups = 0;
  std::string err;
  std::string devinfo;
  // resident training data
  int N = 0, D = 0, npad = 0;
  DevBuf dX;  // N x D
  DevBuf dY;  // N (gpc_nll_batch_cm forms r = y - m0 on the device)
  std::vector<double> hy;
  // workspace
  DevBuf mA, mW, mT;             // matrices of the current chunk
  DevBuf xs, spb, mulb, divb;    // scaled inputs and per-sample scalars
  DevBuf dvec, rvec, zvec, avec; // per-sample vectors (npad each)
  DevBuf tpart;                  // W^T z partial sums
  DevBuf scal;                   // logdet | quad | (ints) info
  DevBuf parts, gout, diagq;     // trace pass
  DevBuf dmb, dsn2b, mg, ng;     // mean / noise gradient inputs and outputs
  DevBuf ks, vb, kss, xss, pout; // predict / predict_full / quad
  DevBuf dbg1, dbg2, dbg3;       // debug hooks / fetch staging
  PinBuf pin;                    // pinned staging for host<->device transfers (see PinBuf)
  // Launch graphs of the device pipeline for small problems (npad <= graph_max_npad, one sample
  // group): a single evaluation at N = 200 is ~20 dependent launches of a few microseconds each,
  // so the launch sequence is captured once per shape and replayed (hipGraphLaunch).
  struct GraphEntry {
    unsigned long long key[4] = {0, 0, 0, 0};
    unsigned long long epoch = 0;
    hipGraphExec_t exec = nullptr;
    unsigned long long last_use = 0;
    double flops = 0;  // algorithmic flops of the captured sequence (host-side bookkeeping)
  };
  static constexpr int NGRAPH = 16;
  GraphEntry graphs[NGRAPH];
  unsigned long long graph_clock = 0;
  int graph_max_npad = 4096;     // GPC_GRAPH_MAX_NPAD (0 disables)
  bool capturing = false;
  DevBuf tile_ctr;               // counters of the persistent GEMM launches, CTR_PER_GROUP per sample group
  static constexpr int CTR_PER_GROUP = 1024;
  // Freed posterior storage, kept for the next gpc_posterior_batch: hipMalloc/hipFree of the
  // multi-GB S x npad^2 factors cost more than computing them (update() in a fit / active-learning
  // loop creates a new posterior set and drops the previous one every iteration).
  std::vector<DevBuf> pool;
  DevBuf pool_take(size_t need) {
    int best = -1;
    for (int i = 0; i < (int)pool.size(); ++i)
      if (pool[i].bytes >= need && pool[i].bytes <= need + need / 2 &&
          (best < 0 || pool[i].bytes < pool[best].bytes))
        best = i;
    DevBuf r;
    if (best >= 0) {
      r = pool[best];
      pool.erase(pool.begin() + best);
    }
    return r;
  }
  void pool_give(DevBuf& b) {
    if (!b.p) return;
    pool.push_back(b);
    b.p = nullptr;
    b.bytes = 0;
    while (pool.size() > 6) {  // two generations of (A, W, alpha)
      pool.front().release();
      pool.erase(pool.begin());
    }
  }
  void pool_drain() {
    for (DevBuf& b : pool) b.release();
    pool.clear();
  }
  double ms_total = 0, ms_factor = 0;
  double last_flops = 0;
  // test hooks (gpc_set_option): first jitter multiplier of every factorization (the reference
  // starts at 1, gaussian_process.py:2402) and samples whose rank-one append is declared unstable
  double start_mult = 1.0;
  unsigned append_fail_mask = 0;
  int retry_runs = 0;  // device pipelines spent on jitter retries by the last call (one per level)
#ifdef GPC_EXPERIMENTS
  // ---- tile-level dataflow (dag.h).  Option "dag": 0 off, 1 wherever the plan supports it, -1 automatic (by what was
  // measured to win).  The arithmetic is that of the stream-ordered schedule bit for bit, so the choice may follow the
  // batch size.
  int dag = 0;
  int dag_small_tiles = 40;  // launches with fewer 128-tiles per sample than this become 64-tile tasks (urgent ring)
  int dag_lauum = 1;         // W^T W inside the graph (its tiles fill the CUs other samples' chains leave idle)
  int dag_leaf_blocks = 0;   // leaf servers; 0: min(samples, 8)
  int dag_gate = 0;          // > 0: sample s starts when sample s - dag_gate has finished the leaf at dag_gate_pct % of its chain
  int dag_gate_pct = 50;
  int dag_urgent_cus = 0;    // > 0: CUs per shader engine (on a team's urgent XCD) whose workgroups serve the urgent ring only
  int dag_crit_pct = 15;     // tasks with less slack than this share of the critical path go to the urgent / crit rings
  int dag_timeout_ms = 2000; // a workgroup that finds nothing to do for this long aborts the graph
  int dag_runs = 0, dag_aborts = 0;  // statistics (gpc_get_option "dag_runs" / "dag_aborts")
  bool dag_used = false;     // the pipeline in flight runs a graph: its abort word rides back with the results
  struct DagEntry {
    unsigned long long key = 0;
    gpc::DagPlan plan;  // (host vectors are released after the upload; the counts stay)
    DevBuf tasks, succ, launches_rel;
    unsigned long long last_use = 0;
  };
  std::vector<DagEntry> dag_cache;
  unsigned long long dag_clock = 0;
  DevBuf dag_pending, dag_slots, dag_ctl, dag_launches;
  DevBuf rsv_tbl1;  // cu_reserve_bail table with ONE CU per XCD (the leaf servers' room)
  DevBuf rsv_tbl4;  // ... with one CU per shader engine (independent pipelines)
  // Independent pipelines for small batches of large problems (option "indep"; round 5): every sample its own stream and
  // pipeline, every chip-filling launch CU-reserving, so that one sample's leaves and small launches run beside another's bulk
  int indep = 0, indep_max = 4, indep_min_tiles = 64;
#endif  // GPC_EXPERIMENTS
};

struct gpc_post {
  gpc_ctx* ctx = nullptr;
  int dtype = GPC_F64;
  int S = 0, N = 0, D = 0, npad = 0;
  CovDesc cd{};
  DevBuf A, W;  // per-sample: A = Cholesky factor (L_chol) or -(K+Sigma)^-1 ; W = L^-1
  DevBuf alpha; // S x npad (double)
  std::vector<double> sp, mul, dv;  // host copies of per-sample scalars (SP_STRIDE), scaling
  std::vector<double> sW, mult;
  std::vector<int> lchol, info;
  // What every predict / quad call of this posterior needs again -- the per-sample scalars, the input scaling and
  // the scaled training inputs -- stays on the device once the first call has put it there (three uploads and a
  // kernel less per call); any change of sp / mul / dv / N clears the flag.
  DevBuf dsp, dmul, ddv, dxs;
  bool dev_consts = false;
};

#define HIPCHK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
      return -1;                                                                            \
    }                                                                                       \
  } while (0)

#define FAIL(ctx, msg)   \
  do {                   \
    (ctx)->err = (msg);  \
    return -2;           \
  } while (0)

namespace {

// Which (XCC_ID, SE_ID, CU_ID) triples does this device run workgroups on?  Enough short-lived blocks to cover every
// CU several times over; each marks the CU it finds itself on and lingers a few microseconds so that the
// dispatcher has to spread the grid.
__global__ __launch_bounds__(256) void cu_probe_kernel(unsigned* __restrict__ seen, long long spin_ticks) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicOr(seen + (xcc & 7) * 8 + ((hw >> 13) & 0x7), 1u << ((hw >> 8) & 0xf));
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) {
  }
}

int probe_cu_map(gpc_ctx* c, int n_cus) {
  HIPCHK(c, c->dbg1.ensure(64 * sizeof(unsigned)));
  HIPCHK(c, hipMemsetAsync(c->dbg1.p, 0, 64 * sizeof(unsigned), c->st));
  int seen_total = 0;
  for (int attempt = 0; attempt < 3 && seen_total < n_cus; ++attempt) {
    hipLaunchKernelGGL(cu_probe_kernel, dim3(16 * std::max(n_cus, 1)), dim3(256), 0, c->st, c->dbg1.as<unsigned>(), 500ll);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->cu_seen, c->dbg1.p, sizeof c->cu_seen, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    seen_total = 0;
    for (unsigned m : c->cu_seen) seen_total += __builtin_popcount(m);
  }
  // usable for reserving CUs: every CU accounted for, and every engine that exists has the same number of them
  int per_engine = -1, engines = 0;
  bool uniform = true;
  for (unsigned m : c->cu_seen) {
    if (!m) continue;
    ++engines;
    const int k = __builtin_popcount(m);
    if (per_engine < 0) per_engine = k;
    uniform = uniform && k == per_engine;
  }
  c->cu_map_ok = seen_total == n_cus && uniform && engines > 0;
  char buf[160];
  snprintf(buf, sizeof buf, " cu_map=%s(%d CUs seen in %d engines of %d)", c->cu_map_ok ? "ok" : "UNRECOGNISED: deferred schedule off",
           seen_total, engines, per_engine);
  c->devinfo += buf;
  return 0;
}

#ifdef GPC_EXPERIMENTS
// XCDs the CU probe saw workgroups on: the dataflow graph serves the urgent ring of team x from XCD x only, so it needs all
// NQ of them (a partitioned device would stall every graph until its time-out: ADVICE r5)
int xcds_seen(const gpc_ctx* c) {
  int n = 0;
  for (int x = 0; x < 8; ++x) {
    unsigned any = 0;
    for (int se = 0; se < 8; ++se) any |= c->cu_seen[x * 8 + se];
    n += any != 0;
  }
  return n;
}
#endif

// rsv_tbl for `reserve` CUs per XCD: the highest-numbered CUs of each shader engine -- 2: one in every other engine,
// 4: one per engine, 8: two, 12: three, ...; >= 32 (test hook): every CU, which leaves the work to the one block the
// survivor rule of cu_reserve_bail keeps.
int build_reserve_table(gpc_ctx* c) {
  unsigned short tbl[64] = {};
  const int r = c->defer_reserve;
  for (int x = 0; x < 8; ++x) {
    int engine_no = 0;
    for (int se = 0; se < 8; ++se) {
      unsigned m = c->cu_seen[x * 8 + se];
      if (!m) continue;
      int k = r >= 32 ? 16 : (r == 2 ? (engine_no % 2 == 0 ? 1 : 0) : r / 4);
      ++engine_no;
      unsigned out = 0;
      for (int cu = 15; cu >= 0 && k > 0; --cu)
        if ((m >> cu) & 1u) {
          out |= 1u << cu;
          --k;
        }
      tbl[x * 8 + se] = (unsigned short)out;
    }
  }
  HIPCHK(c, c->rsv_tbl.ensure(sizeof tbl));
  HIPCHK(c, hipMemcpyAsync(c->rsv_tbl.p, tbl, sizeof tbl, hipMemcpyHostToDevice, c->st));
#ifdef GPC_EXPERIMENTS
  // the dataflow schedule (dag.h) keeps ONE CU per XCD free of GEMM workgroups for its leaf servers: the
  // highest-numbered CU of the first shader engine seen in each XCD
  unsigned short tbl1[64] = {};
  for (int x = 0; x < 8; ++x)
    for (int se = 0; se < 8; ++se) {
      const unsigned m = c->cu_seen[x * 8 + se];
      if (!m) continue;
      tbl1[x * 8 + se] = (unsigned short)(1u << (31 - __builtin_clz(m)));
      break;
    }
  HIPCHK(c, c->rsv_tbl1.ensure(sizeof tbl1));
  HIPCHK(c, hipMemcpyAsync(c->rsv_tbl1.p, tbl1, sizeof tbl1, hipMemcpyHostToDevice, c->st));
  unsigned short tbl4[64] = {};  // the highest-numbered CU of EVERY shader engine (every engine can drain its share of a grid)
  for (int i = 0; i < 64; ++i)
    if (c->cu_seen[i]) tbl4[i] = (unsigned short)(1u << (31 - __builtin_clz(c->cu_seen[i])));
  HIPCHK(c, c->rsv_tbl4.ensure(sizeof tbl4));
  HIPCHK(c, hipMemcpyAsync(c->rsv_tbl4.p, tbl4, sizeof tbl4, hipMemcpyHostToDevice, c->st));
#endif
  HIPCHK(c, hipStreamSynchronize(c->st));
  return 0;
}

int cov_count_of(int kind, int D) {
  switch (kind) {
    case K_SE:
    case K_MATERN:
      return D + 1;
    case K_RQ:
      return D + 2;
    case K_SE_ISO:
    case K_MATERN_ISO:
      return 2;
    default:
      return -1;
  }
}

bool valid_kernel(int kind, int degree) {
  if (kind < K_SE || kind > K_MATERN_ISO) return false;
  if ((kind == K_MATERN || kind == K_MATERN_ISO) && degree != 1 && degree != 3 && degree != 5) return false;
  return true;
}

// input scaling exactly as the reference applies it before the distance computation
void scaling_of(int kind, int degree, int D, const double* hyp, double* mul, double* dv, double* sf2,
                double* rqa) {
  const double snu = std::sqrt((double)degree);
  *rqa = 1.0;
  if (cov_is_iso(kind)) {
    const double ell = std::exp(hyp[0]);
    *sf2 = std::exp(2 * hyp[1]);
    for (int h = 0; h < D; ++h) {
      mul[h] = (kind == K_MATERN_ISO) ? snu : 1.0;  // X * sqrt(nu) / ell   |  X / ell
      dv[h] = ell;
    }
    return;
  }
  *sf2 = std::exp(2 * hyp[D]);
  if (kind == K_RQ) *rqa = std::exp(hyp[D + 1]);
  for (int h = 0; h < D; ++h) {
    const double ell = std::exp(hyp[h]);
    if (kind == K_SE) {
      mul[h] = 1.0;  // X / ell
      dv[h] = ell;
    } else if (kind == K_MATERN) {
      mul[h] = snu / ell;  // X @ diag(sqrt(nu)/ell)
      dv[h] = 1.0;
    } else {
      mul[h] = 1.0 / ell;  // X @ diag(1/ell)
      dv[h] = 1.0;
    }
  }
}

// host-side description of a batch of hyperparameter samples
struct Batch {
  int S = 0, N = 0, D = 0, npad = 0;
  CovDesc cd{};
  bool vec_noise = false;
  bool m_const = false;  // m holds ONE value per sample (constant / zero mean: gpc_nll_batch_cm); r is formed on the device
  const double* hyp_cov = nullptr;
  const double* m = nullptr;
  const double* sn2 = nullptr;
  const double* y = nullptr;
  std::vector<double> mul, dv, sp, dvec, r;
  std::vector<double> smin, mult, sl;
  std::vector<int> lchol, tries, info;
  double mult0 = 1.0;  // first jitter multiplier (1 in the reference; a test hook can raise it)

  void init() {
    mul.assign((size_t)S * D, 1.0);
    dv.assign((size_t)S * D, 1.0);
    sp.assign((size_t)S * SP_STRIDE, 0.0);
    dvec.assign((size_t)S * npad, 1.0);
    if (!m_const) r.assign((size_t)S * npad, 0.0);
    smin.assign(S, 0.0);
    mult.assign(S, mult0);
    sl.assign(S, 1.0);
    lchol.assign(S, 0);
    tries.assign(S, 0);
    info.assign(S, 0);
    for (int s = 0; s < S; ++s) {
      double sf2 = 0.0, rqa = 1.0;
      if (hyp_cov)  // built-in kernel; with a caller-provided K (cd.kind < 0) there is nothing to scale
        scaling_of(cd.kind, cd.degree, D, hyp_cov + (size_t)s * cd.cov_N, &mul[(size_t)s * D],
                   &dv[(size_t)s * D], &sf2, &rqa);
      sp[(size_t)s * SP_STRIDE + SP_SF2] = sf2;
      sp[(size_t)s * SP_STRIDE + SP_RQA] = rqa;
      const double* sn = sn2 + (size_t)s * (vec_noise ? N : 1);
      double mn = sn[0];
      if (vec_noise)
        for (int i = 1; i < N; ++i) mn = std::min(mn, sn[i]);
      smin[s] = mn;
      lchol[s] = (mn >= 1e-6) ? 1 : 0;  // gaussian_process.py:2404
      if (!m_const)
        for (int i = 0; i < N; ++i) r[(size_t)s * npad + i] = y[i] - m[(size_t)s * N + i];
      apply_mult(s);
    }
  }

  // (re)derive the scaled system of sample s for its current jitter multiplier
  // (gaussian_process.py:2406-2439)
  void apply_mult(int s) {
    const double* sn = sn2 + (size_t)s * (vec_noise ? N : 1);
    double* d = &dvec[(size_t)s * npad];
    if (lchol[s]) {
      const double sn2_div = smin[s];  // scalar noise: the value itself
      sl[s] = sn2_div * mult[s];
      for (int i = 0; i < N; ++i) d[i] = vec_noise ? sn[i] / sn2_div : 1.0;
    } else {
      sl[s] = 1.0;
      for (int i = 0; i < N; ++i) d[i] = mult[s] * (vec_noise ? sn[i] : sn[0]);
    }
    for (int i = N; i < npad; ++i) d[i] = 1.0;
    sp[(size_t)s * SP_STRIDE + SP_KSCALE] = sl[s];  // K is DIVIDED by this
    sp[(size_t)s * SP_STRIDE + SP_SL] = sl[s];
  }
};


enum Mode { MODE_NLL = 0, MODE_GRAD = 1, MODE_POST = 2 };

// GPC_HOSTTIME=1: host wall-clock of the phases of a call, to stderr (diagnostics only)
struct HostClock {
  bool on;
  const char* tag;
  std::chrono::steady_clock::time_point t;
  explicit HostClock(const char* tg) : on(getenv("GPC_HOSTTIME") != nullptr), tag(tg) {
    if (on) t = std::chrono::steady_clock::now();
  }
  void lap(const char* what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[gpc %s] %-10s %8.3f ms\n", tag, what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

template <typename T>
struct Pipe {
  gpc_ctx* c;
  Batch* B;
  Mode mode;
  // matrices for the chunk being processed (sample 0 of the chunk)
  T* A;
  T* W;
  T* Tm;
  long long sM;
  // gradient side inputs (whole batch, host pointers)
  const double* dm = nullptr;
  int mean_N = 0;
  const double* dsn2 = nullptr;
  int noise_N = 0;
  // outputs (whole batch, host)
  std::vector<double> logdet, quad, G, mg, ng;
  // caller-provided covariance ("K-mode", gpc_nll_batch_K / gpc_posterior_batch_K): host pointer to the
  // N x N matrix of every sample of the batch, and the callback that delivers dK one plane at a time
  std::vector<const double*> Kptr;
  gpc_dk_plane_fn dk_cb = nullptr;
  void* dk_user = nullptr;
  std::vector<int> orig;  // index of each sample in the caller's numbering (for the callback)
  bool kmode() const { return !Kptr.empty(); }

  int P() const { return B->cd.cov_N + 1; }

  // K-mode: sum_ij Q_ij dK_ij/dtheta_p for the sample in workspace slot `slot`, one plane at a time
  // (the caller's compute() produced dK on the host; only one N x N plane is ever resident here).
  int trace_planes(int s, int slot) {
    Batch& b = *B;
    const int N = b.N, npad = b.npad, cov_N = b.cd.cov_N, Pn = P();
    hipStream_t st = c->st;
    const size_t pb = (size_t)N * N * sizeof(double);
    HIPCHK(c, c->dbg1.ensure(pb));
    const int nblk = (int)std::min<long long>(1024, ((long long)N * N + 255) / 256);
    HIPCHK(c, c->dbg2.ensure((size_t)nblk * sizeof(double) + 64));
    std::vector<double> plane((size_t)N * N);
    for (int p = 0; p < cov_N; ++p) {
      if (dk_cb(dk_user, orig.empty() ? s : orig[s], p, plane.data()) != 0) FAIL(c, "the dK callback reported an error");
      HIPCHK(c, hipMemcpyAsync(c->dbg1.p, plane.data(), pb, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL((trace_plane_kernel<T>), dim3(nblk), dim3(256), 0, st, (const T*)(Tm + (size_t)slot * sM), npad,
                         (const double*)(c->avec.as<double>() + (size_t)slot * npad),
                         (const double*)(c->spb.as<double>() + (size_t)slot * SP_STRIDE), c->dbg1.as<double>(), N,
                         c->dbg2.as<double>());
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(1, 1), dim3(256), 0, st, (const double*)c->dbg2.p, nblk, 1,
                         c->dbg2.as<double>() + nblk);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(&G[(size_t)s * Pn + p], c->dbg2.as<double>() + nblk, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipStreamSynchronize(st));  // `plane` is reused for the next hyperparameter
    }
    return 0;
  }

  // Device kernels for `n` samples of the current chunk starting at chunk index `off`
  // (all per-sample buffers are indexed by chunk position), issued on stream `st`.
  int device_section(hipStream_t st, int off, int n, hipEvent_t f0, hipEvent_t f1, int gidx = gpc_ctx::MAXG) {
    Batch& b = *B;
    const int npad = b.npad, N = b.N, D = b.D;
    T* Ac = A + (size_t)off * sM;
    T* Wc = W + (size_t)off * sM;
    T* Tc = Tm + (size_t)off * sM;
    double* xs = c->xs.as<double>() + (size_t)off * npad * D;
    double* spb = c->spb.as<double>() + (size_t)off * SP_STRIDE;
    double* rvec = c->rvec.as<double>() + (size_t)off * npad;
    double* zvec = c->zvec.as<double>() + (size_t)off * npad;
    double* avec = c->avec.as<double>() + (size_t)off * npad;
    double* d_logdet = c->scal.as<double>() + off;
    double* d_quad = c->scal.as<double>() + chunk_cnt + off;
    int* d_info = reinterpret_cast<int*>(c->scal.as<double>() + 2 * chunk_cnt) + off;
    const int t64 = npad / CT, ntl = t64 * (t64 + 1) / 2;
    split_build = false;
    if (kmode()) {
      // A = K / (sn2_div * sn2_mult) + diag from the caller's matrix, one sample at a time through one
      // N x N staging buffer (stream order keeps the upload of sample i+1 behind the kernel reading i)
      HIPCHK(c, c->dbg1.ensure((size_t)N * N * sizeof(double)));
      for (int i = 0; i < n; ++i) {
        HIPCHK(c, hipMemcpyAsync(c->dbg1.p, Kptr[chunk_s0 + off + i], (size_t)N * N * sizeof(double),
                                 hipMemcpyHostToDevice, st));
        dim3 gl(npad / 64, npad / 4), bl(64, 4);
        hipLaunchKernelGGL((load_K_kernel<T>), gl, bl, 0, st, (const double*)c->dbg1.p, N, npad,
                           (const double*)(spb + (size_t)i * SP_STRIDE),
                           (const double*)(c->dvec.as<double>() + (size_t)(off + i) * npad), Ac + (size_t)i * sM);
      }
    } else {
      if (!prescaled) {
        const long long tot = (long long)npad * D;
        dim3 grid((unsigned)((tot + 255) / 256), n);
        hipLaunchKernelGGL(scale_x_kernel, grid, dim3(256), 0, st, c->dX.as<double>(), N, npad, D,
                           c->mulb.as<double>() + (size_t)off * D, c->divb.as<double>() + (size_t)off * D, xs);
      }
      // With the deferred schedule the build is split: the tiles of the first 1024 rows on this stream, the
      // rest as a persistent, CU-reserving launch on the side stream under the first subtree (plan.h waits
      // for it before the first launch that reads rows >= 1024).
      const int head64 = 1024 / CT, ntl_head = head64 * (head64 + 1) / 2;
      split_build = (defer_node > 0 || use_rl) && !c->capturing && gpc::g_persist_spare >= 0 && npad >= 2048 && reserve_tbl();
      const double* dv = c->dvec.as<double>() + (size_t)off * npad;
      if (split_build) {
        GPC_COV_DISPATCH(build_kernel, T, b.cd, dim3(ntl_head, n), dim3(256), 0, st, b.cd, (const double*)xs,
                         (const double*)spb, dv, N, npad, Ac, sM, npad, 0);
        hipStream_t sd = c->sst[gidx];
        int* bctr = c->tile_ctr.as<int>() + (size_t)gidx * gpc_ctx::CTR_PER_GROUP + gpc_ctx::CTR_PER_GROUP - CTR_STRIDE;
        HIPCHK(c, hipMemsetAsync(bctr, 0, CTR_STRIDE * sizeof(int), st));
        HIPCHK(c, hipEventRecord(c->ev_bfork[gidx], st));
        HIPCHK(c, hipStreamWaitEvent(sd, c->ev_bfork[gidx], 0));
        GPC_COV_DISPATCH(build_persist_kernel, T, b.cd, dim3(4 * gpc::g_block_slots / 2 + 128), dim3(256), 0, sd, b.cd,
                         (const double*)xs, (const double*)spb, dv, N, npad, Ac, sM, npad, ntl_head, ntl, n,
                         reserve_tbl(), bctr);
        HIPCHK(c, hipEventRecord(c->ev_btail[gidx], sd));
      } else {
        GPC_COV_DISPATCH(build_kernel, T, b.cd, dim3(ntl, n), dim3(256), 0, st, b.cd, (const double*)xs,
                         (const double*)spb, dv, N, npad, Ac, sM, npad, 0);
      }
    }
    HIPCHK(c, hipGetLastError());

    if (f0 && !c->capturing) HIPCHK(c, hipEventRecord(f0, st));
    Factor<T> F;
    F.st = st;
    F.batch = n;
    F.npad = npad;
    F.A = Ac;
    F.W = Wc;
    F.Tm = Tc;
    F.sA = F.sW = F.sT = sM;
    F.logdet = d_logdet;
    F.info = d_info;
    F.nvalid = N;
    F.stable = stable || c->stable;
    if ((defer_node > 0 || use_rl) && !c->capturing && gpc::g_persist_spare >= 0) {
      F.side = c->sst[gidx];
      F.evs = c->dev_ev[gidx];
      F.nev = gpc_ctx::NDEV - 2;  // (the last two belong to the solves beside W^T W)
      F.defer_min = defer_node;
      F.reserve = reserve_tbl();
    }
#ifdef GPC_EXPERIMENTS
    if (indep_mode && !c->capturing) {
      F.reserve = c->rsv_tbl4.template as<unsigned short>();
      F.reserved_small_bt = 64;
      F.reserve_all = true;
      F.reserve_min = c->indep_min_tiles;
    }
#endif
    if (c->check_queues && !c->capturing) F.qlog = &qlog;
    if (split_build) {
      F.ev_tail = c->ev_btail[gidx];
      F.tail_row0 = 1024;
    }
    if (gpc::g_persist_spare >= 0) {
      // (the last counters of the group belong to the split build, already in flight on the side stream)
      F.ctr = c->tile_ctr.as<int>() + (size_t)gidx * gpc_ctx::CTR_PER_GROUP;
      F.ctr_cap = gpc_ctx::CTR_PER_GROUP - CTR_STRIDE;
      // (no launch of a small problem is persistent -- its largest, W^T W, has tm (tm + 1) / 2 tiles per sample --
      // and the zeroing would be a launch of its own on the critical path)
      const long long tmx = npad / TILE, biggest = tmx * (tmx + 1) / 2 * n;
      if (defer_node > 0 || F.reserve_all || biggest >= std::min<long long>(gpc::g_small_launch_blocks, gpc::g_block_slots - gpc::g_persist_spare))
        HIPCHK(c, hipMemsetAsync(F.ctr, 0, (gpc_ctx::CTR_PER_GROUP - CTR_STRIDE) * sizeof(int), st));
      else
        F.ctr = nullptr;
    }
    // NLL only: the inverse of the whole matrix is not needed (only of left children)
    const bool full_inv = (mode != MODE_NLL);
    const int nll_blk = c->nll_block >= 0 ? c->nll_block
                                          : (npad >= 2048 ? 512 : 0);
    const bool nll_blocked = !use_rl && mode == MODE_NLL && nll_blk >= TILE && npad > nll_blk;
    bool dag_done = false, dag_has_lauum = false;
    if (nll_blocked) F.nll_block = nll_blk;  // (the blocked forward solve below reads it, whichever schedule factors)
#ifdef GPC_EXPERIMENTS
    if (use_dag && !use_rl) {
      int rc = dag_section(st, gidx, F, nll_blocked ? nll_blk : 0, full_inv, dag_has_lauum);
      if (rc < 0) return rc;
      dag_done = rc == 0;  // (rc > 0: the plan is not one the graph executes -- the stream-ordered schedule below)
      if (dag_done && getenv("GPC_DAG_SYNC")) {
        (void)hipStreamSynchronize(c->sst[gidx]);
        (void)hipStreamSynchronize(st);
      }
    }
    if (!dag_done && use_rl) {
      F.rl_panel = c->rl_panel;
      F.rl_lookahead = (double)n * std::pow((double)npad / 4096.0, 3.0) <= (double)c->rl_ahead_max;
      F.potrf_rl();
      dag_done = true;  // (the factorization has been issued)
    }
#endif
    if (dag_done) {
      // (experiments build: the factorization -- and W^T W when it is part of the graph -- has been issued)
    } else if (nll_blocked) {
      F.nll_block = nll_blk;
      F.potrf_nll(0, npad);
    } else {
      F.potrf_inv(0, npad, full_inv, mode == MODE_POST);
    }
    // z = L^-1 r ; quad = z.z ; alpha = W^T z / sl -- two triangular matrix-vector products over W, HBM-bound and
    // independent of W^T W.  With the gradient they run UNDER the W^T W launch (round 3): its two resident blocks per
    // CU use 2 x 240 of the 512 VGPRs per lane and 147 of 160 KB of LDS, so kernels of at most 32 VGPRs (trmv_low_kernel,
    // trmv_t_part_low_kernel, blas1.h) are dispatched beside them and take their bytes while the GEMM is
    // MFMA-bound; forked on the side stream just before the launch, joined before the gradient contraction, which
    // needs alpha.  (Eager one- or multi-group pipelines; a captured graph keeps the serial order.)
    // (fp64 only: the fp32 GEMM uses 249 VGPRs, two of its blocks leave no register for anything else)
    const bool solves_beside_lauum = mode == MODE_GRAD && !c->capturing && !kmode() && c->solves_beside_lauum &&
                                     sizeof(T) == 8 && npad >= 2048 && gpc::g_persist_spare >= 0 && !dag_has_lauum && !indep_mode;
    auto solves = [&](hipStream_t sx) -> int {
      const hipStream_t keep = F.st;
      F.st = sx;
      F.low_regs = sx != st;
#ifdef GPC_EXPERIMENTS
      if (use_rl)
        F.forward_solve_rl(rvec, zvec);
      else
#endif
      if (nll_blocked)
        F.forward_solve_nll(0, npad, rvec, zvec);
      else
        F.forward_solve(0, npad, full_inv, rvec, zvec);
      F.st = keep;
      F.low_regs = false;
      if (mode == MODE_NLL) {
        hipLaunchKernelGGL(dot_kernel, dim3(1, n), dim3(256), 0, sx, (const double*)zvec, (const double*)zvec,
                           npad, npad, d_quad);
      } else {  // quad = z.z rides in the first block of the transposed product
        double* tpart = c->tpart.as<double>() + (size_t)off * (npad / TRC) * npad;
        const dim3 gt((npad + 64 * MM<T>::VEC - 1) / (64 * MM<T>::VEC), npad / TRC, n);
        if (sx != st)
          hipLaunchKernelGGL((trmv_t_part_low_kernel<T>), gt, dim3(256), 0, sx, (const T*)Wc, sM, npad, (const double*)zvec,
                             npad, tpart, d_quad);
        else
          hipLaunchKernelGGL((trmv_t_part_kernel<T, 8>), gt, dim3(256), 0, sx, (const T*)Wc, sM, npad, (const double*)zvec,
                             npad, tpart, d_quad);
        hipLaunchKernelGGL(trmv_t_sum_kernel, dim3(npad / 128, n), dim3(128), 0, sx, (const double*)tpart, npad,
                           (const double*)spb, (int)SP_STRIDE, (int)SP_SL, avec);
      }
      HIPCHK(c, hipGetLastError());
      return 0;
    };
    hipEvent_t ev_solved = nullptr;
    if (mode == MODE_GRAD) {
      if (solves_beside_lauum) {
        hipStream_t sx = c->sst[gidx];
        hipEvent_t ev_fork = c->dev_ev[gidx][gpc_ctx::NDEV - 1];
        ev_solved = c->dev_ev[gidx][gpc_ctx::NDEV - 2];
        HIPCHK(c, hipEventRecord(ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(sx, ev_fork, 0));
        if (int rc = solves(sx)) return rc;
        HIPCHK(c, hipEventRecord(ev_solved, sx));
      }
      if (!dag_has_lauum) {
        if (!c->capturing) HIPCHK(c, hipEventRecord(c->ev_l0[gidx], st));
        F.lauum(Tc, sM);
        if (!c->capturing) {
          HIPCHK(c, hipEventRecord(c->ev_l1[gidx], st));
          lauum_n[gidx] = n;
        }
      }
    }
    HIPCHK(c, F.err);
    HIPCHK(c, hipGetLastError());
    if (f1 && !c->capturing) HIPCHK(c, hipEventRecord(f1, st));
    c->last_flops += F.flops;

    if (ev_solved)
      HIPCHK(c, hipStreamWaitEvent(st, ev_solved, 0));
    else if (int rc = solves(st))
      return rc;
    HIPCHK(c, hipGetLastError());

    const int Pn = P();
    if (mode == MODE_GRAD) {
      double* parts = c->parts.as<double>() + (size_t)off * ntl * Pn;
      double* diagq = c->diagq.as<double>() + (size_t)off * npad;
      if (kmode()) {
        // diag(Q) and trace(Q) now; the covariance slots are filled plane by plane after the sync (trace_planes)
        hipLaunchKernelGGL((diagq_kernel<T>), dim3(1, n), dim3(256), 0, st, (const T*)Tc, sM, npad, (const double*)avec,
                           (const double*)spb, N, diagq, c->gout.as<double>() + (size_t)off * Pn, Pn);
      } else {
        GPC_COV_DISPATCH(trace_kernel, T, b.cd, dim3(ntl, n), dim3(256), 4 * (((Pn + 3) & ~3) + 4) * sizeof(double), st,
                         b.cd, (const double*)xs, (const double*)spb, (const double*)avec, N, npad, (const T*)Tc, sM,
                         npad, parts, ntl, diagq);
        // the reduction of the tile partials and the mean / noise gradient products: one launch
        const int mN = mean_N > 0 ? mean_N : 0, nN = (noise_N > 0 && b.vec_noise) ? noise_N : 0;
        hipLaunchKernelGGL(grad_tail_kernel, dim3(Pn + mN + nN, n), dim3(256), 0, st, (const double*)parts, ntl, Pn,
                           c->gout.as<double>() + (size_t)off * Pn,
                           (mN && dm) ? (const double*)(c->dmb.as<double>() + (size_t)off * N * mean_N) : nullptr, N, mN,
                           (const double*)avec, mN ? c->mg.as<double>() + (size_t)off * mean_N : nullptr,
                           nN ? (const double*)(c->dsn2b.as<double>() + (size_t)off * N * noise_N) : nullptr, nN,
                           (const double*)diagq, nN ? c->ng.as<double>() + (size_t)off * noise_N : nullptr, npad);
        HIPCHK(c, hipGetLastError());
        return 0;
      }
      if (mean_N > 0)
        hipLaunchKernelGGL(mat_t_vec_kernel, dim3(mean_N, n), dim3(256), 0, st,
                           (const double*)(c->dmb.as<double>() + (size_t)off * N * mean_N), N, mean_N,
                           (const double*)avec, npad, c->mg.as<double>() + (size_t)off * mean_N);
      if (noise_N > 0 && b.vec_noise)
        hipLaunchKernelGGL(mat_t_vec_kernel, dim3(noise_N, n), dim3(256), 0, st,
                           (const double*)(c->dsn2b.as<double>() + (size_t)off * N * noise_N), N, noise_N,
                           (const double*)diagq, npad, c->ng.as<double>() + (size_t)off * noise_N);
      HIPCHK(c, hipGetLastError());
    }
    return 0;
  }

#ifdef GPC_EXPERIMENTS
  // ---- tile-level dataflow (dag.h) ----------------------------------------------------------------------------
  bool use_dag = false;  // run(): this pipeline's factorization goes through the task graph
  bool indep_mode = false;  // run(): one pipeline per sample, chip-filling launches CU-reserving
  int dag_trace_n = 0;
  // "dag" = -1: where the graph was measured to win
  bool dag_auto(int cnt, int npad) const { return false; }

  // The graph of (npad, plan variant): recorded from plan.h with placeholder buffer addresses, cut into tile tasks,
  // uploaded once and kept.  Returns nullptr when the plan holds something the graph does not execute.
  gpc_ctx::DagEntry* dag_plan(int nll_blk, bool full_inv, bool with_lauum, double* flops1) {
    Batch& b = *B;
    const int npad = b.npad;
    const unsigned long long key = (unsigned long long)npad | ((unsigned long long)sizeof(T) << 20) |
                                   ((unsigned long long)nll_blk << 24) | ((unsigned long long)full_inv << 40) |
                                   ((unsigned long long)with_lauum << 41) | ((unsigned long long)mode << 42) |
                                   ((unsigned long long)c->dag_small_tiles << 44) | ((unsigned long long)c->dag_crit_pct << 56);
    for (auto& e : c->dag_cache)
      if (e.key == key) {
        e.last_use = ++c->dag_clock;
        *flops1 = e.plan.work_us;  // (work_us is reused to carry the algorithmic flops of one sample, see below)
        return e.plan.ntasks > 0 ? &e : nullptr;
      }
    PlanRecorder rec;
    Factor<T> F;
    F.rec = &rec;
    F.st = nullptr;
    F.batch = 1;
    F.npad = npad;
    F.A = static_cast<T*>(const_cast<void*>(dag_fake_base(0)));
    F.W = static_cast<T*>(const_cast<void*>(dag_fake_base(1)));
    F.Tm = static_cast<T*>(const_cast<void*>(dag_fake_base(2)));
    F.sA = F.sW = F.sT = sM;
    F.nvalid = b.N;
    F.dual_launch = false;
    if (nll_blk > 0) {
      F.nll_block = nll_blk;
      F.potrf_nll(0, npad);
    } else {
      F.potrf_inv(0, npad, full_inv, false);
    }
    if (with_lauum) F.lauum(F.Tm, sM);
    if (c->dag_cache.size() >= 8) {  // least recently used entry goes
      size_t old = 0;
      for (size_t i = 1; i < c->dag_cache.size(); ++i)
        if (c->dag_cache[i].last_use < c->dag_cache[old].last_use) old = i;
      c->dag_cache[old].tasks.release();
      c->dag_cache[old].succ.release();
      c->dag_cache[old].launches_rel.release();
      c->dag_cache.erase(c->dag_cache.begin() + old);
    }
    c->dag_cache.emplace_back();
    gpc_ctx::DagEntry& e = c->dag_cache.back();
    e.key = key;
    e.last_use = ++c->dag_clock;
    const void* bases[3] = {dag_fake_base(0), dag_fake_base(1), dag_fake_base(2)};
    DagPlan& P = e.plan;
    if (!build_dag(rec, bases, npad, sizeof(T), c->dag_small_tiles, P, false, c->dag_crit_pct / 100.0)) {
      P = DagPlan{};
      P.work_us = F.flops;
      *flops1 = F.flops;
      return nullptr;
    }
    if (getenv("GPC_DAG_LOG"))
      fprintf(stderr, "[gpcore] dag npad=%d nll_blk=%d lauum=%d: %d tasks (%d leaves, %d x 64, %d x 128), %lld edges, %zu launches; "
                      "rings: %d urgent, %d crit, %d bulk; model: critical path %.0f us, work %.0f CU-us\n", npad, nll_blk, (int)with_lauum, P.ntasks, P.nleaf, P.n64, P.n128,
              P.nedges, P.launches.size(), P.n_urgent, P.n_crit, P.n_bulk, P.crit_us, P.work_us);
    auto put = [&](DevBuf& d, const void* src, size_t n) -> hipError_t {
      hipError_t er = d.ensure(std::max<size_t>(n, 16));
      if (er != hipSuccess) return er;
      return hipMemcpy(d.p, src, n, hipMemcpyHostToDevice);
    };
    if (put(e.tasks, P.tasks.data(), P.tasks.size() * sizeof(DagTask)) != hipSuccess ||
        put(e.succ, P.succ.data(), P.succ.size() * sizeof(int)) != hipSuccess ||
        put(e.launches_rel, P.launches.data(), P.launches.size() * sizeof(GemmArgs)) != hipSuccess) {
      c->dag_cache.pop_back();
      return nullptr;
    }
    P.work_us = F.flops;  // from here on: algorithmic flops of one sample (tile-exact, as plan.h counts them)
    std::vector<DagTask>().swap(P.tasks);
    std::vector<int>().swap(P.succ);
    *flops1 = F.flops;
    return &e;
  }

  // Issues the graph for the n samples of F.  0: issued; > 0: not applicable (the caller falls back); < 0: error.
  int dag_section(hipStream_t st, int gidx, Factor<T>& F, int nll_blk, bool full_inv, bool& has_lauum) {
    const int n = F.batch, npad = F.npad;
    const bool with_lauum = mode == MODE_GRAD && c->dag_lauum != 0;
    double flops1 = 0;
    gpc_ctx::DagEntry* e = dag_plan(nll_blk, full_inv, with_lauum, &flops1);
    if (!e) return 1;
    const DagPlan& P = e->plan;
    const int nlaunch = (int)P.launches.size();
    // ring storage: urgent | NQ bulk rings | leaf
    DagDev d{};
    int off = 0;
    d.nteams = std::min(n, (int)NQ);
    for (int q = 0; q < NQ; ++q) {  // team q holds the samples q, q + nteams, ...
      const int members = q < d.nteams ? (n - q + d.nteams - 1) / d.nteams : 0;
      d.ring_base[DAG_RING_URGENT + q] = off;
      off += members * P.n_urgent;
      d.ring_base[DAG_RING_CRIT0 + q] = off;
      off += members * P.n_crit;
      d.ring_base[DAG_RING_BULK0 + q] = off;
      off += members * P.n_bulk;
    }
    // staggered starts: with more samples than teams the chains of all samples would otherwise run in phase (every
    // sample in its latency-bound stretches at the same time, nothing to fill the chip with)
    d.gate = c->dag_gate > 0 && c->dag_gate < n ? c->dag_gate : 0;
    d.gate_task = P.leaf_task.empty() ? 0 : P.leaf_task[std::min<size_t>(P.leaf_task.size() - 1,
                                                                       (size_t)(P.leaf_task.size() * (size_t)c->dag_gate_pct / 100))];
    d.ring_base[DAG_RING_LEAF] = off;
    off += n * P.nleaf;
    HIPCHK(c, c->dag_slots.ensure((size_t)std::max(off, 4) * sizeof(int)));
    HIPCHK(c, c->dag_pending.ensure((size_t)n * P.ntasks * sizeof(int)));
    HIPCHK(c, c->dag_ctl.ensure(sizeof(DagCtl)));
    HIPCHK(c, c->dag_launches.ensure((size_t)std::max(nlaunch, 1) * sizeof(GemmArgs)));
    d.tasks = e->tasks.template as<DagTask>();
    d.succ = e->succ.template as<int>();
    d.launches_rel = e->launches_rel.template as<GemmArgs>();
    d.launches_w = c->dag_launches.template as<GemmArgs>();
    d.launches = d.launches_w;
    d.nlaunch = nlaunch;
    d.base[0] = F.A;
    d.base[1] = F.W;
    d.base[2] = F.Tm;
    d.pending = c->dag_pending.template as<int>();
    d.slots = c->dag_slots.template as<int>();
    d.ctl = c->dag_ctl.template as<DagCtl>();
    d.ntasks = P.ntasks;
    d.S = n;
    d.A = F.A;
    d.W = F.W;
    d.sA = F.sA;
    d.sW = F.sW;
    d.npad = npad;
    d.nvalid = F.nvalid;
    d.logdet = F.logdet;
    d.info = F.info;
    d.rsv = c->rsv_tbl1.template as<unsigned short>();
    d.timeout_ticks = (long long)c->dag_timeout_ms * 100000ll;  // wall_clock64: 100 MHz
    d.trace = nullptr;
    if (getenv("GPC_DAG_TRACE")) {
      HIPCHK(c, c->dbg2.ensure((size_t)n * P.ntasks * 6 * sizeof(long long)));
      HIPCHK(c, hipMemsetAsync(c->dbg2.p, 0, (size_t)n * P.ntasks * 6 * sizeof(long long), st));
      d.trace = c->dbg2.template as<long long>();
      dag_trace_n = n * P.ntasks;
    }
    HIPCHK(c, hipMemsetAsync(d.slots, 0, (size_t)off * sizeof(int), st));
    HIPCHK(c, hipMemsetAsync(d.ctl, 0, sizeof(DagCtl), st));
    const long long items = (long long)n * P.ntasks;
    hipLaunchKernelGGL(dag_init_kernel, dim3((unsigned)std::min<long long>(1024, (items + 255) / 256)), dim3(256), 0, st, d);
    // The leaf servers run on the pipeline's own stream (created with the greatest priority: they are the latency-bound
    // chain), the GEMM workers on the side stream (least priority, like every chip-filling side launch of plan.h).
    hipStream_t sd = c->sst[gidx];
    hipEvent_t ev_fork = c->dev_ev[gidx][0], ev_join = c->dev_ev[gidx][1];
    HIPCHK(c, hipEventRecord(ev_fork, st));
    HIPCHK(c, hipStreamWaitEvent(sd, ev_fork, 0));
    if (mode == MODE_GRAD && with_lauum) HIPCHK(c, hipEventRecord(c->ev_l0[gidx], sd));
    // EXACTLY two workgroups per CU (74 KB of LDS each: the grid fills every CU by pigeonhole); the two that land on the
    // reserved CU of each XCD return at once and leave it empty for a leaf server.  No surplus: workgroups that cannot
    // be placed stay pending in the dispatcher, and a dispatch with pending workgroups kept the YOUNGER leaf launch from
    // starting at all (measured: an oversized grid drained only through the shader engine that holds the reserved CU,
    // 24 of 80 surplus workgroups; the leaf servers started 12 us after the workers had given up).
    d.urgent_cus = c->dag_urgent_cus;
    d.leaf_servers = c->dag_leaf_blocks >= 0 ? std::max(1, std::min(c->dag_leaf_blocks > 0 ? c->dag_leaf_blocks : std::min(n, 8), 8)) : 0;
    hipLaunchKernelGGL((dag_worker_kernel<T>), dim3(gpc::g_block_slots), dim3(256), 0, sd, d);
    if (mode == MODE_GRAD && with_lauum) {
      HIPCHK(c, hipEventRecord(c->ev_l1[gidx], sd));
      lauum_n[gidx] = -n;  // negative: the timed launch is the whole graph (run() prices it as S N^3 flops)
    }
    HIPCHK(c, hipEventRecord(ev_join, sd));
    // (test hook: dag_leaf_blocks < 0 starts NO leaf server -- the graph stalls at its first leaf, the workers' bounded
    // waits run out, the graph aborts and run() falls back to the stream-ordered schedule)
    const int nl = std::max(1, std::min(c->dag_leaf_blocks > 0 ? c->dag_leaf_blocks : std::min(n, 8), 8));
    if (c->dag_leaf_blocks >= 0)
      hipLaunchKernelGGL((dag_leaf_kernel<T>), dim3(nl), dim3(256), 0, st, d, gpc::g_leaf_fault);
    HIPCHK(c, hipStreamWaitEvent(st, ev_join, 0));
    HIPCHK(c, hipGetLastError());
    F.flops += flops1 * n;
    F.launches += 2;
    has_lauum = with_lauum;
    c->dag_used = true;
    ++c->dag_runs;
    return 0;
  }
#else
  // (the product library runs the stream-ordered schedule only: the conditions below fold away)
  static constexpr bool use_dag = false, indep_mode = false, use_rl = false;
#endif
  // device_section through a cached launch graph (one sample group, main stream)
  int graph_section(int cnt) {
    Batch& b = *B;
    hipStream_t st = c->st;
    const unsigned long long key[4] = {
        ((unsigned long long)mode << 60) | ((unsigned long long)sizeof(T) << 52) | ((unsigned long long)b.vec_noise << 48) |
            ((unsigned long long)prescaled << 49) | ((unsigned long long)(stable || c->stable) << 50) |
            ((unsigned long long)cnt << 24) | (unsigned long long)b.N,
        ((unsigned long long)b.cd.kind << 48) | ((unsigned long long)b.cd.degree << 40) | ((unsigned long long)b.D << 20) |
            ((unsigned long long)mean_N << 10) | (unsigned long long)noise_N,
        (unsigned long long)(uintptr_t)A, (unsigned long long)(uintptr_t)W ^ ((unsigned long long)(uintptr_t)Tm << 1)};
    gpc_ctx::GraphEntry* hit = nullptr;
    gpc_ctx::GraphEntry* victim = &c->graphs[0];
    for (auto& g : c->graphs) {
      if (g.exec && g.epoch == g_alloc_epoch && g.key[0] == key[0] && g.key[1] == key[1] && g.key[2] == key[2] &&
          g.key[3] == key[3])
        hit = &g;
      if (g.last_use < victim->last_use) victim = &g;
    }
    if (!hit) {
      if (victim->exec) {
        (void)hipGraphExecDestroy(victim->exec);
        victim->exec = nullptr;
      }
      hipGraph_t graph = nullptr;
      const double flops_before = c->last_flops;
      HIPCHK(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      c->capturing = true;
      const int rc = device_section(st, 0, cnt, nullptr, nullptr);
      c->capturing = false;
      victim->flops = c->last_flops - flops_before;
      const hipError_t e = hipStreamEndCapture(st, &graph);
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      HIPCHK(c, e);
      const hipError_t ei = hipGraphInstantiate(&victim->exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      HIPCHK(c, ei);
      for (int i = 0; i < 4; ++i) victim->key[i] = key[i];
      victim->epoch = g_alloc_epoch;
      hit = victim;
    } else {
      c->last_flops += hit->flops;
    }
    hit->last_use = ++c->graph_clock;
    if (timing_on) HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipGraphLaunch(hit->exec, st));
    if (timing_on) HIPCHK(c, hipEventRecord(c->ev[2], st));
    return 0;
  }

  // the table of CUs a deferred / split launch stays off, or nullptr when the device's CU numbering was not
  // recognised or no CU is to be reserved (the launches are then plain persistent ones)
  const unsigned short* reserve_tbl() const {
    return (c->cu_map_ok && c->defer_reserve > 0) ? c->rsv_tbl.as<unsigned short>() : nullptr;
  }
  std::vector<typename Factor<T>::QueueCheck> qlog;  // debug option check_queues

  // check_queues: every tile queue of every persistent launch of the pipeline that just ran must have handed out all
  // its tiles (a queue counter below its total = tiles nobody computed: the grid was starved)
  int verify_queues() {
    if (qlog.empty()) return 0;
    std::vector<int> h((gpc_ctx::MAXG + 1) * gpc_ctx::CTR_PER_GROUP);
    HIPCHK(c, hipMemcpy(h.data(), c->tile_ctr.p, h.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (const auto& q : qlog) {
      const int* v = h.data() + (q.slot - c->tile_ctr.as<int>());
      long long drawn = 0;
      const bool affine = (gpc::g_gemm_flags & 8) != 0;
      if (!affine) {
        drawn = std::min<long long>(v[0], (long long)q.ntiles * q.batch);
      } else {
        const int nclass = q.batch >= NQ ? 1 : NQ / q.batch;
        for (int k = 0; k < NQ; ++k) {
          long long total;
          if (q.batch >= NQ)
            total = (long long)q.ntiles * ((q.batch - k + NQ - 1) / NQ);
          else
            total = k >= q.batch * nclass ? 0 : (q.ntiles - k / q.batch + nclass - 1) / nclass;
          drawn += std::min<long long>(v[k], total);
        }
      }
      if (drawn != (long long)q.ntiles * q.batch) {
        qlog.clear();
        c->err = "internal error: a persistent GEMM launch ended with tiles left in its queues";
        return -3;
      }
    }
    qlog.clear();
    return 0;
  }

  int chunk_cnt = 0;
  int chunk_s0 = 0;  // first sample (batch numbering) of the chunk being processed
  int defer_node = 0;  // plan.h: nodes at least this large run their U product on the side stream (0: none)
  bool split_build = false;
#ifdef GPC_EXPERIMENTS
  bool use_rl = false;  // plan.h: right-looking panels with look-ahead for this pipeline
#endif
  bool stable = false;  // plan.h: refined panel solves (jitter retries; gpc_set_option "stable")
  bool prescaled = false;  // the transfer kernel of this chunk has written the scaled inputs (run())
  bool timing_on = true;   // run_once(): this call records its timing events (always from N_pad = 2048 on; below on request)
  std::vector<double> r_expanded;  // run(): r = y - m0 formed on the host when no pinned block was to be had
  int lauum_n[gpc_ctx::MAXG + 1] = {};

  // The pipeline of a chunk whose matrices are single 128 x 128 leaves (see run()).  A, W, Tm point at the chunk's slot.
  int small_section(int s0, int cnt, const double* hsp, const double* hmul, const double* hdv, const double* hdvec,
                    const double* rsrc, size_t scal_bytes, HostClock& hc) {
    Batch& b = *B;
    const int npad = b.npad, N = b.N, D = b.D, Pn = P();
    hipStream_t st = c->st;
    double* d_logdet = c->scal.as<double>();
    double* d_quad = d_logdet + cnt;
    int* d_info = reinterpret_cast<int*>(d_quad + cnt);
    // (without gradient -- an evaluation, or a posterior: L in A, W and alpha are what it keeps -- the three scalars of a
    // sample are all that comes back)
    // (the landing block: the context's coherent block when the results fit and polling is on -- its last word is the
    // completion flag --, else a piece of the staging block and a stream synchronisation)
    // (a gradient evaluation ends with the gathered download launch instead, which announces itself the same way:
    // XferDesc::flag)
    const bool poll = c->small_poll && c->land_blk && scal_bytes + 16 <= gpc_ctx::LAND_BYTES && cnt <= 64;
    double* land = mode != MODE_GRAD ? (poll ? c->land_blk : static_cast<double*>(c->pin.alloc(scal_bytes))) : nullptr;
    unsigned long long* flag = poll ? reinterpret_cast<unsigned long long*>(c->land_blk + (gpc_ctx::LAND_BYTES / 8 - 1)) : nullptr;
    const unsigned long long seq = ++c->land_seq;
    int* done_ctr = reinterpret_cast<int*>(reinterpret_cast<char*>(c->scal.p) + scal_bytes);  // zeroed with the scalars
    const bool timing = c->small_timing != 0;
    XferDesc u;
    HIPCHK(c, c->pin.take_up(st, c->scal.p, scal_bytes + 8, u));
    u.X = c->dX.as<double>();
    u.mul = hmul;
    u.dv = hdv;
    u.xs = c->xs.as<double>();
    u.n = N;
    u.npad = npad;
    u.D = D;
    u.cnt = cnt;
    if (timing) HIPCHK(c, hipEventRecord(c->ev[1], st));
    GPC_COV_DISPATCH(small_front_kernel, T, b.cd, dim3(3, cnt), dim3(256), 0, st, u, b.cd, hsp, hdvec, b.vec_noise ? 1 : 0,
                     A, sM);
    hipLaunchKernelGGL((leaf_solve_kernel<T>), dim3(cnt), dim3(256), 0, st, A, sM, npad, W, sM, npad, d_logdet, d_info, N,
                       gpc::g_leaf_fault, rsrc, c->zvec.as<double>(), d_quad,
                       mode != MODE_NLL ? c->avec.as<double>() : nullptr, (const double*)c->spb.as<double>(),
                       (int)SP_STRIDE, (int)SP_SL, land, cnt, mode != MODE_GRAD ? flag : nullptr, seq, done_ctr);
    HIPCHK(c, hipGetLastError());
    c->last_flops += (2.0 / 3.0) * TILE * (double)TILE * TILE * cnt;
    if (mode == MODE_GRAD) {
      Factor<T> F;
      F.st = st;
      F.batch = cnt;
      F.npad = npad;
      F.A = A;
      F.W = W;
      F.Tm = Tm;
      F.sA = F.sW = F.sT = sM;
      F.lauum(Tm, sM);
      HIPCHK(c, F.err);
      c->last_flops += F.flops;
      HIPCHK(c, hipEventRecord(c->ev[2], st));
      const int t64 = npad / CT, ntl = t64 * (t64 + 1) / 2;
      double* parts = c->parts.as<double>();
      double* diagq = c->diagq.as<double>();
      GPC_COV_DISPATCH(trace_kernel, T, b.cd, dim3(ntl, cnt), dim3(256), 4 * (((Pn + 3) & ~3) + 4) * sizeof(double), st, b.cd,
                       (const double*)c->xs.as<double>(), (const double*)c->spb.as<double>(),
                       (const double*)c->avec.as<double>(), N, npad, (const T*)Tm, sM, npad, parts, ntl, diagq);
      const int mN = mean_N > 0 ? mean_N : 0, nN = (noise_N > 0 && b.vec_noise) ? noise_N : 0;
      hipLaunchKernelGGL(grad_tail_kernel, dim3(Pn + mN + nN, cnt), dim3(256), 0, st, (const double*)parts, ntl, Pn,
                         c->gout.as<double>(), (mN && dm) ? (const double*)c->dmb.as<double>() : nullptr, N, mN,
                         (const double*)c->avec.as<double>(), mN ? c->mg.as<double>() : nullptr,
                         nN ? (const double*)c->dsn2b.as<double>() : nullptr, nN, (const double*)diagq,
                         nN ? c->ng.as<double>() : nullptr, npad);
      HIPCHK(c, hipGetLastError());
    } else if (timing) {
      HIPCHK(c, hipEventRecord(c->ev[2], st));
    }
    if (timing) HIPCHK(c, hipEventRecord(c->ev[3], st));
    hc.lap("launch");
    std::vector<double> hscal(scal_bytes / 8);
    if (mode == MODE_GRAD) {
      HIPCHK(c, c->pin.gather(hscal.data(), d_logdet, scal_bytes, st));
      HIPCHK(c, c->pin.gather(&G[(size_t)s0 * Pn], c->gout.p, (size_t)cnt * Pn * 8, st));
      if (mean_N > 0) HIPCHK(c, c->pin.gather(&mg[(size_t)s0 * mean_N], c->mg.p, (size_t)cnt * mean_N * 8, st));
      if (noise_N > 0 && b.vec_noise)
        HIPCHK(c, c->pin.gather(&ng[(size_t)s0 * noise_N], c->ng.p, (size_t)cnt * noise_N * 8, st));
      const bool fl = poll && !c->pin.down_plain;
      HIPCHK(c, c->pin.flush_down(st, fl ? flag : nullptr, seq, done_ctr));
      if (!fl) flag = nullptr;
    } else if (!land) {
      HIPCHK(c, hipMemcpyAsync(hscal.data(), d_logdet, scal_bytes, hipMemcpyDeviceToHost, st));
    }
    bool seen = false;
    if (poll && flag && !timing) {
      // (bounded: a call that has not announced itself after 150 us -- a large batch, a busy device -- is waited for the
      // ordinary way, which also covers a device that never writes the word)
      const volatile unsigned long long* fw = flag;
      const auto t_poll = std::chrono::steady_clock::now();
      for (int spin = 0; !seen; ++spin) {
        if (*fw == seq) {
          seen = true;
          break;
        }
        if ((spin & 63) == 63 &&
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_poll).count() > 150.0)
          break;
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (seen) {
      ++c->small_polled;  // every store of the call is visible (the flag was its last); the stream drains by itself
    } else {
      HIPCHK(c, hipStreamSynchronize(st));
      ++c->small_synced;
    }
    c->pin.finish();
    if (land) memcpy(hscal.data(), land, scal_bytes);
    memcpy(&logdet[s0], hscal.data(), (size_t)cnt * 8);
    memcpy(&quad[s0], hscal.data() + cnt, (size_t)cnt * 8);
    if (getenv("GPC_SCALAR_LOG"))
      for (int i = 0; i < cnt; ++i)
        fprintf(stderr, "[gpcore] sample %d logdet %.17g quad %.17g\n", s0 + i, logdet[s0 + i], quad[s0 + i]);
    const int* hinfo = reinterpret_cast<const int*>(hscal.data() + 2 * (size_t)cnt);
    hc.lap("d2h+sync");
    for (int i = 0; i < cnt; ++i) b.info[s0 + i] = hinfo[i];
    for (int i = 0; i < cnt; ++i)
      if (hinfo[i] & LEAF_TIMEOUT) {
        c->err = "internal error: a hand-off between the waves of a 128 x 128 leaf factorization timed out (sample " +
                 std::to_string(s0 + i) + "); the results of this call are invalid";
        return -3;
      }
    if (timing) {
      float t03 = 0, t12 = 0;
      (void)hipEventElapsedTime(&t03, c->ev[0], c->ev[3]);
      (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
      c->ms_total += t03;
      c->ms_factor += t12;
    }
    return 0;
  }

  // run the device pipeline for samples [s0, s0+cnt) whose matrices start at slot `slot`.
  // The chunk is split into sample groups on separate HIP streams: the latency-bound
  // phases of one group (leaves, deep recursion levels) run beside the throughput-bound
  // GEMMs of the other.
  // One pipeline for the samples [s0, s0 + cnt).  A dataflow graph that aborted (a bounded wait ran out: never
  // expected) is not an error of the call: the batch goes again on the stream-ordered schedule.
#ifdef GPC_EXPERIMENTS
  static constexpr int DAG_RETRY = 77;
  bool dag_off_once = false;
  int run(int s0, int cnt, int slot) {
    int rc = run_once(s0, cnt, slot);
    if (rc == DAG_RETRY) {
      ++c->dag_aborts;
      if (getenv("GPC_DAG_LOG")) fprintf(stderr, "[gpcore] dataflow graph aborted; re-running on the stream-ordered schedule\n");
      dag_off_once = true;
      rc = run_once(s0, cnt, slot);
      dag_off_once = false;
    }
    return rc;
  }
#else
  int run(int s0, int cnt, int slot) { return run_once(s0, cnt, slot); }
#endif

  int run_once(int s0, int cnt, int slot) {
    Batch& b = *B;
    const int npad = b.npad, N = b.N, D = b.D;
    hipStream_t st = c->st;
    const size_t vb = (size_t)npad * sizeof(double);
    const int t64 = npad / CT, ntl = t64 * (t64 + 1) / 2;
    const int Pn = P();
    chunk_cnt = cnt;
    chunk_s0 = s0;
    // the pipe's matrix pointers are relative to chunk position 0
    T* A0 = A;
    T* W0 = W;
    T* T0 = Tm;
    A = A0 + (size_t)slot * sM;
    W = W0 + (size_t)slot * sM;
    Tm = T0 + (size_t)slot * sM;
    struct Restore {
      Pipe* p;
      T *a, *w, *t;
      ~Restore() { p->A = a; p->W = w; p->Tm = t; }
    } restore{this, A0, W0, T0};

    HIPCHK(c, c->xs.ensure((size_t)cnt * npad * D * sizeof(double)));
    HIPCHK(c, c->spb.ensure((size_t)cnt * SP_STRIDE * sizeof(double)));
    HIPCHK(c, c->mulb.ensure((size_t)cnt * D * sizeof(double)));
    HIPCHK(c, c->divb.ensure((size_t)cnt * D * sizeof(double)));
    HIPCHK(c, c->dvec.ensure(cnt * vb));
    HIPCHK(c, c->rvec.ensure(cnt * vb));
    HIPCHK(c, c->zvec.ensure(cnt * vb));
    HIPCHK(c, c->avec.ensure(cnt * vb));
    HIPCHK(c, c->tpart.ensure((size_t)cnt * (npad / TRC + 1) * vb));
    HIPCHK(c, c->scal.ensure((size_t)cnt * (2 * sizeof(double) + sizeof(int)) + 64));
    double* d_logdet = c->scal.as<double>();
    double* d_quad = d_logdet + cnt;
    int* d_info = reinterpret_cast<int*>(d_quad + cnt);
    if (mode == MODE_GRAD) {
      HIPCHK(c, c->parts.ensure((size_t)cnt * ntl * Pn * sizeof(double)));
      HIPCHK(c, c->gout.ensure((size_t)cnt * Pn * sizeof(double)));
      HIPCHK(c, c->diagq.ensure(cnt * vb));
      if (mean_N > 0) {
        if (dm) HIPCHK(c, c->dmb.ensure((size_t)cnt * N * mean_N * 8));
        HIPCHK(c, c->mg.ensure((size_t)cnt * mean_N * 8));
      }
      if (noise_N > 0 && b.vec_noise) {
        HIPCHK(c, c->dsn2b.ensure((size_t)cnt * N * noise_N * 8));
        HIPCHK(c, c->ng.ensure((size_t)cnt * noise_N * 8));
      }
    }

    HostClock hc("run");
    c->pin.begin();
    c->pin.begin_gather();
    auto up = [&](void* dst, const void* src, size_t n) { return c->pin.stage(dst, src, n); };
    // (a one-leaf evaluation records its timing events only on request: each is a barrier packet on a 45 us pipeline)
    const bool small_cand = c->small_path && npad == TILE && !kmode() && !(stable || c->stable) && gpc::g_leaf_version == 5;
    // ... and so does every problem below N_pad = 2048, whose call is 0.1 - 1 ms of dependent launches (the timing of the
    // larger ones feeds bench.py's roofline figures and costs them nothing measurable)
    timing_on = c->small_timing != 0 || npad >= 2048;
    bool ev0_pending = small_cand && !c->small_timing;
    if (timing_on && !ev0_pending) HIPCHK(c, hipEventRecord(c->ev[0], st));
    const void* hsp = up(c->spb.p, &b.sp[(size_t)s0 * SP_STRIDE], (size_t)cnt * SP_STRIDE * 8);
    const void* hmul = up(c->mulb.p, &b.mul[(size_t)s0 * D], (size_t)cnt * D * 8);
    const void* hdv = up(c->divb.p, &b.dv[(size_t)s0 * D], (size_t)cnt * D * 8);
    // the scaled inputs of the whole chunk ride in the transfer kernel (device_section then starts with the build)
    prescaled = hmul && hdv && !kmode() && (size_t)cnt * npad * D * 8 <= (PinBuf::kGather << 2);
    if (prescaled) {
      XferDesc& u = c->pin.upd;
      u.X = c->dX.as<double>();
      u.mul = static_cast<const double*>(hmul);
      u.dv = static_cast<const double*>(hdv);
      u.xs = c->xs.as<double>();
      u.n = N;
      u.npad = npad;
      u.D = D;
      u.cnt = cnt;
    }
    // Problems that are ONE 128 x 128 leaf take a pipeline of their own (small_section below); there the diagonal term
    // is read from its staged host copy by the build itself and needs no device copy
    const bool small_ok = small_cand && hsp && hmul && hdv;
    // The diagonal term and r = y - m.  Scalar noise / a constant mean (gpc_nll_batch_cm) are ONE value per sample:
    // they are staged as such and expanded on the device by the upload kernel (no S x N arrays cross the bus).
    const void* hdvec = nullptr;
    const void* hr = nullptr;
    bool fill_dvec = false, fill_r = false;
    XferDesc& fu = c->pin.upd;
    if (!b.vec_noise && !kmode()) {
      if (void* h = c->pin.alloc((size_t)cnt * 8)) {
        for (int i = 0; i < cnt; ++i) static_cast<double*>(h)[i] = b.dvec[(size_t)(s0 + i) * npad];
        if (small_ok) {
          hdvec = h;  // read by the build of the one-leaf pipeline directly
        } else {
          fu.dval = static_cast<const double*>(h);
          fu.dvec_out = c->dvec.as<double>();
          fill_dvec = true;
        }
      }
    } else if (small_ok) {
      if (void* h = c->pin.alloc(cnt * vb)) {
        memcpy(h, &b.dvec[(size_t)s0 * npad], cnt * vb);
        hdvec = h;
      }
    }
    if (!hdvec && !fill_dvec) up(c->dvec.p, &b.dvec[(size_t)s0 * npad], cnt * vb);
    if (b.m_const) {
      if (void* h = c->pin.alloc((size_t)cnt * 8)) {
        memcpy(h, b.m + s0, (size_t)cnt * 8);
        fu.y = c->dY.as<double>();
        fu.m0 = static_cast<const double*>(h);
        fu.r_out = c->rvec.as<double>();
        fill_r = true;
      } else {
        // no pinned block for the per-sample constants: r = y - m0 (the same subtraction, so the same bits) is expanded
        // here and travels like every other staged input, with its plain-copy fallback
        r_expanded.assign((size_t)cnt * npad, 0.0);
        for (int i = 0; i < cnt; ++i)
          for (int k = 0; k < N; ++k) r_expanded[(size_t)i * npad + k] = b.y[k] - b.m[s0 + i];
        up(c->rvec.p, r_expanded.data(), cnt * vb);
      }
    } else {
      // r = y - m is read from its staged copy by the kernel that needs it (leaf_solve_kernel, at its start: the round
      // trip to host memory hides under the factorization)
      if (small_ok && hdvec)
        if (void* h = c->pin.alloc(cnt * vb)) {
          memcpy(h, &b.r[(size_t)s0 * npad], cnt * vb);
          hr = h;
        }
      if (!hr) up(c->rvec.p, &b.r[(size_t)s0 * npad], cnt * vb);
    }
    if (fill_dvec || fill_r) {
      fu.fn = N;
      fu.fnpad = npad;
      fu.fcnt = cnt;
    }
    // (dm == nullptr with mean_N = 1: the constant mean's derivative, all ones -- nothing to upload)
    if (mode == MODE_GRAD && mean_N > 0 && dm) up(c->dmb.p, dm + (size_t)s0 * N * mean_N, (size_t)cnt * N * mean_N * 8);
    if (mode == MODE_GRAD && noise_N > 0 && b.vec_noise)
      up(c->dsn2b.p, dsn2 + (size_t)s0 * N * noise_N, (size_t)cnt * N * noise_N * 8);
    // Problems that are ONE 128 x 128 leaf (N <= 128; decided by the problem size only, so that a row of a batch carries
    // the bits of its single evaluation): upload + build in one launch, leaf + both triangular products in one launch,
    // and an evaluation without gradient writes its three scalars straight into the host's landing block.  Two
    // launches instead of six (NLL), six instead of ten (gradient).  gpc_set_option("small_path", 0) turns it off.
    const size_t scal_bytes0 = (((size_t)cnt * (2 * sizeof(double) + sizeof(int))) + 7) & ~(size_t)7;
    if (small_ok && hdvec) {
      int rc = small_section(s0, cnt, static_cast<const double*>(hsp), static_cast<const double*>(hmul),
                             static_cast<const double*>(hdv), static_cast<const double*>(hdvec),
                             hr ? static_cast<const double*>(hr) : (const double*)c->rvec.as<double>(), scal_bytes0, hc);
      return rc;
    }
    // one kernel: every staged segment and the zeroing of [logdet | quad | info] (padded to whole words)
    if (timing_on && ev0_pending) HIPCHK(c, hipEventRecord(c->ev[0], st));  // (a one-leaf candidate that takes the general pipeline after all)
    // (+ 8 bytes: the counter of the download launch's completion flag, see XferDesc)
    HIPCHK(c, c->pin.flush_up(st, c->scal.p, scal_bytes0 + 8));

    hc.lap("h2d");
    int groups = c->groups;
    if (cnt < 2 * groups || npad < 1024 || kmode()) groups = 1;
    // Two sample groups on two streams pay for their half-size launches only when the batch is large: measured with
    // the final kernels (tools/batch_latency.py), one group is faster up to S (npad/4096)^3 = 64 in both modes --
    // N=1000 S=8: 0.78 -> 0.54 ms (NLL), 0.92 -> 0.68 ms (gradient); N=2000 S=8 NLL 1.78 -> 1.36 ms; cfg3 NLL-only
    // 9.44 -> 9.04 ms -- and two groups by 1-2 % beyond (N=4096 S=128).
    if ((double)cnt * std::pow((double)npad / 4096.0, 3.0) <= 64.0) groups = 1;
    // Deferred inverse products (plan.h): measured on MI355X (round-2 sweep, script since removed) they pay whenever the
    // latency-bound phases are a visible share of the batch -- S (npad/4096)^3 <= 64 with at least 4 samples:
    // N=2048 S=16 4.42 -> 4.04 ms, N=4096 S=4 8.59 -> 7.47, N=4096 S=16 21.5 -> 20.2, N=4096 S=32 39.4 -> 38.7
    // -- and cost a little beyond (N=8192 S=64: 556 -> 584 ms).  With them one sample group is better than two.
    defer_node = 0;
    if (mode != MODE_NLL && c->defer_min != 0 && (c->cu_map_ok || c->defer_min > 0)) {
      const double work = (double)cnt * std::pow((double)npad / 4096.0, 3.0);
      if (c->defer_min > 0)
        defer_node = c->defer_min;
      else if (cnt >= 4 && npad >= 2048 && work <= 64.0)
        defer_node = (npad / 2 / TILE) * TILE;
      // one or two samples of a large problem: the right child of the root is latency-bound for a long stretch
      // (top two levels, as above: N=8192 S=1 12.95 -> 12.2 ms, S=2 20.87 -> 19.97 ms; N=16384 fp32 S=1 40.0 -> 38.9 ms;
      // N=4096 S=1 loses 2 %, so not below work = 8)
      else if (cnt < 4 && work >= 8.0 && work <= 64.0)
        defer_node = (npad / 2 / TILE) * TILE;
      if (defer_node > 0) groups = 1;
    }
#ifdef GPC_EXPERIMENTS
    // right-looking panels with look-ahead (NLL-only evaluations; side stream + CU reservation: eager)
    use_rl = mode == MODE_NLL && c->rl_panel >= TILE && npad >= 4 * c->rl_panel && !(stable || c->stable) && !kmode();
    // Launch graphs: every one-group pipeline whose schedule lives on ONE stream (the deferred products and the split
    // build fork to side streams with CU-reserving launches and stay eager).  Round 3: up to npad = 4096 (round 2
    // stopped at 1024) -- N = 2000: 1.171 -> 1.137 ms, N = 4096: 3.03 -> 2.97 ms per single NLL+grad evaluation.
    // Tile-level dataflow (dag.h) instead of one launch per product: same tile arithmetic, so the choice may follow
    // the batch size.  One sample group, no deferred launches (the graph overlaps them by itself), eager.
    use_dag = false;
    c->dag_used = false;
    if (c->dag != 0 && !dag_off_once && !kmode() && !(stable || c->stable) && mode != MODE_POST && !use_rl &&
        gpc::g_leaf_version == 5 && c->cu_map_ok && xcds_seen(c) == NQ && npad >= 2 * TILE && c->dag_aborts < 3) {
      use_dag = c->dag > 0 || dag_auto(cnt, npad);
      if (use_dag) {
        groups = 1;
        defer_node = 0;
      }
    }
    // Independent pipelines (option "indep"): a batch of 2 .. indep_max samples of a large problem as one pipeline PER
    // SAMPLE on its own stream, every chip-filling launch persistent and off one CU per shader engine -- the leaves and small
    // launches of one sample then run on those CUs while another sample's products fill the rest (lock-step runs every
    // sample's chain at the same time and nothing beside it).  Same tiles, same bits.
    indep_mode = c->indep != 0 && !use_dag && !use_rl && !kmode() && !(stable || c->stable) && mode != MODE_POST && c->cu_map_ok &&
                 cnt >= 2 && cnt <= c->indep_max && npad >= 2048 && gpc::g_persist_spare >= 0;
    if (indep_mode) {
      groups = cnt;
      defer_node = 0;
    }
#endif
    // Launch graphs: every one-group pipeline whose schedule lives on ONE stream (the deferred products and the split
    // build fork to side streams with CU-reserving launches and stay eager), up to npad = 4096.
    if (groups == 1 && defer_node == 0 && !use_rl && !use_dag && c->graph_max_npad > 0 && npad <= c->graph_max_npad && !kmode()) {
      int rc = graph_section(cnt);
      if (rc) return rc;
    } else if (groups == 1) {
      int rc = device_section(st, 0, cnt, timing_on ? c->ev[1] : nullptr, timing_on ? c->ev[2] : nullptr);
      if (rc) return rc;
    } else {
      HIPCHK(c, hipEventRecord(c->ev_up, st));
      for (int g = 0; g < groups; ++g) {
        const int lo = (int)((long long)cnt * g / groups), hi = (int)((long long)cnt * (g + 1) / groups);
        hipStream_t sg = c->gst[g];
        HIPCHK(c, hipStreamWaitEvent(sg, c->ev_up, 0));
        int rc = device_section(sg, lo, hi - lo, (g == 0 && timing_on) ? c->ev[1] : nullptr, nullptr, g);
        if (rc) {  // the groups already launched still read this call's buffers: let them drain
          for (int h = 0; h <= g; ++h) (void)hipStreamSynchronize(c->gst[h]);
          (void)hipStreamSynchronize(st);
          return rc;
        }
        HIPCHK(c, hipEventRecord(c->ev_done[g], sg));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_done[g], 0));
      }
      if (timing_on) HIPCHK(c, hipEventRecord(c->ev[2], st));
    }
    if (timing_on) HIPCHK(c, hipEventRecord(c->ev[3], st));

    hc.lap("launch");
    // results back
    std::vector<int> hinfo((cnt + 1) & ~1);
    // [logdet | quad | info] are contiguous on the device: one segment (info padded to a whole word)
    const size_t scal_bytes = (((size_t)cnt * (2 * sizeof(double) + sizeof(int))) + 7) & ~(size_t)7;
    std::vector<double> hscal(scal_bytes / 8);
    HIPCHK(c, c->pin.gather(hscal.data(), d_logdet, scal_bytes, st));
    if (mode == MODE_GRAD) {
      HIPCHK(c, c->pin.gather(&G[(size_t)s0 * Pn], c->gout.p, (size_t)cnt * Pn * 8, st));
      if (mean_N > 0) HIPCHK(c, c->pin.gather(&mg[(size_t)s0 * mean_N], c->mg.p, (size_t)cnt * mean_N * 8, st));
      if (noise_N > 0 && b.vec_noise)
        HIPCHK(c, c->pin.gather(&ng[(size_t)s0 * noise_N], c->ng.p, (size_t)cnt * noise_N * 8, st));
    }
#ifdef GPC_EXPERIMENTS
    int dag_words[2] = {0, 0};  // [abort, leaf servers started] of the graph, when one ran
    if (c->dag_used)
      HIPCHK(c, c->pin.gather(dag_words, reinterpret_cast<char*>(c->dag_ctl.p) + offsetof(DagCtl, abort), 8, st));
#endif
    // The call ends with the gathered download launch.  Below N_pad = 2048 the host watches the word that launch writes
    // last (XferDesc::flag) instead of the stream: the runtime's completion path costs several microseconds that a call
    // of 0.1 - 1 ms notices.  Bounded: after 2 ms without the word the stream is waited for the ordinary way.
    const bool poll = !timing_on && c->small_poll && c->land_blk && !kmode() && !c->pin.down_plain && !c->check_queues &&
                      c->pin.downd.nseg > 0;
    unsigned long long* flag = poll ? reinterpret_cast<unsigned long long*>(c->land_blk + (gpc_ctx::LAND_BYTES / 8 - 1)) : nullptr;
    const unsigned long long seq = ++c->land_seq;
    HIPCHK(c, c->pin.flush_down(st, flag, seq, reinterpret_cast<int*>(reinterpret_cast<char*>(c->scal.p) + scal_bytes0)));
    bool seen = false;
    if (poll) {
      const volatile unsigned long long* fw = flag;
      const auto t_poll = std::chrono::steady_clock::now();
      for (int spin = 0; !seen; ++spin) {
        if (*fw == seq) {
          seen = true;
          break;
        }
        if ((spin & 63) == 63 &&
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_poll).count() > 2000.0)
          break;
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (seen) {
      ++c->small_polled;
    } else {
      HIPCHK(c, hipStreamSynchronize(st));
      if (poll) ++c->small_synced;
    }
    c->pin.finish();
#ifdef GPC_EXPERIMENTS
    if (c->dag_used && dag_trace_n > 0 && getenv("GPC_DAG_TRACE")) {  // every (sample, task)'s times, one line each, to the named file
      std::vector<long long> tr((size_t)dag_trace_n * 6);
      (void)hipMemcpy(tr.data(), c->dbg2.p, tr.size() * sizeof(long long), hipMemcpyDeviceToHost);
      if (FILE* f = fopen(getenv("GPC_DAG_TRACE"), "w")) {
        for (int t = 0; t < dag_trace_n; ++t)
          fprintf(f, "%d %lld %lld %lld %lld %lld %lld\n", t, tr[(size_t)t * 6], tr[(size_t)t * 6 + 1], tr[(size_t)t * 6 + 2],
                  tr[(size_t)t * 6 + 3], tr[(size_t)t * 6 + 4], tr[(size_t)t * 6 + 5]);
        fclose(f);
      }
      dag_trace_n = 0;
    }
#ifdef GPC_TILE_TRACE
    if (c->dag_used && getenv("GPC_DAG_STATS")) {
      unsigned long long tt[2][4];
      (void)hipMemcpyFromSymbol(tt, HIP_SYMBOL(gpc::g_tile_trace), sizeof tt);
      unsigned long long zero[2][4] = {};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(gpc::g_tile_trace), zero, sizeof zero);
      for (int q = 0; q < 2; ++q)
        if (tt[q][3])
          fprintf(stderr, "[gpcore] %s-tile tasks: %llu; us per task (thread 0): prologue %.2f, k-loop %.2f, epilogue (loads of C, stores issued) %.2f\n",
                  q ? "64" : "128", tt[q][3], tt[q][0] / 100.0 / tt[q][3], tt[q][1] / 100.0 / tt[q][3], tt[q][2] / 100.0 / tt[q][3]);
    }
#endif
    if (c->dag_used && getenv("GPC_DAG_STATS")) {
      DagCtl h;
      (void)hipMemcpy(&h, c->dag_ctl.p, sizeof h, hipMemcpyDeviceToHost);
      const double w = std::max<double>(1.0, (double)h.n_workers), k = 1e-3 / 100.0;  // clock64: 100 MHz constant clock -> us
      fprintf(stderr, "[gpcore] dag stats: %llu workers, %llu tasks (%llu kept); per worker us: life %.0f = pop %.0f + acquire %.0f + run %.0f + complete %.0f; "
                      "per task us: pop %.2f acquire %.2f run %.2f complete %.2f\n", h.n_workers, h.n_tasks, h.n_kept, h.t_life / w * k * 1e3,
              h.t_pop / w * k * 1e3, h.t_acq / w * k * 1e3, h.t_exec / w * k * 1e3, h.t_done / w * k * 1e3,
              h.t_pop * k * 1e3 / std::max<double>(1.0, (double)h.n_tasks), h.t_acq * k * 1e3 / std::max<double>(1.0, (double)h.n_tasks),
              h.t_exec * k * 1e3 / std::max<double>(1.0, (double)h.n_tasks), h.t_done * k * 1e3 / std::max<double>(1.0, (double)h.n_tasks));
    }
    if (c->dag_used && dag_words[0] != 0) {
      if (getenv("GPC_DAG_LOG")) {
        DagCtl h;
        (void)hipMemcpy(&h, c->dag_ctl.p, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[gpcore] dag abort code %d, leaf servers started %d, remaining %d, reserve started %d survivors %d; rings (head/tail):",
                h.abort, h.leaf_alive, h.remaining, h.reserve_ctr[CTR_STARTED], h.reserve_ctr[CTR_SURVIVORS]);
        for (int r = 0; r < DAG_NRINGS; ++r) fprintf(stderr, " %d/%d", *h.ring[r].head(), *h.ring[r].tail());
        fprintf(stderr, "; us: leaf server start %d, first worker start %d, first worker exit %d; leaf servers saw (head tail remaining | exit us):",
                h.pad[0], h.pad[1], h.pad[2]);
        for (int q = 0; q < 4; ++q) fprintf(stderr, " [%d %d %d | %d]", h.pad[4 + 4 * q], h.pad[5 + 4 * q], h.pad[6 + 4 * q], h.pad[7 + 4 * q]);
        fprintf(stderr, "\n");
      }
      for (int g = 0; g <= gpc_ctx::MAXG; ++g) lauum_n[g] = 0;
      return DAG_RETRY;
    }
#endif
    memcpy(&logdet[s0], hscal.data(), (size_t)cnt * 8);
    memcpy(&quad[s0], hscal.data() + cnt, (size_t)cnt * 8);
    if (getenv("GPC_SCALAR_LOG"))
      for (int i = 0; i < cnt; ++i)
        fprintf(stderr, "[gpcore] sample %d logdet %.17g quad %.17g\n", s0 + i, logdet[s0 + i], quad[s0 + i]);
    memcpy(hinfo.data(), hscal.data() + 2 * (size_t)cnt, (size_t)cnt * sizeof(int));
    hc.lap("d2h+sync");
    for (int i = 0; i < cnt; ++i) b.info[s0 + i] = hinfo[i];
    for (int i = 0; i < cnt; ++i)
      if (hinfo[i] & LEAF_TIMEOUT) {
        c->err = "internal error: a hand-off between the waves of a 128 x 128 leaf factorization timed out (sample " +
                 std::to_string(s0 + i) + "); the results of this call are invalid";
        return -3;
      }
    if (c->check_queues) {
      int rc = verify_queues();
      if (rc) return rc;
    }
    if (kmode() && mode == MODE_GRAD)
      for (int i = 0; i < cnt; ++i)
        if (hinfo[i] == 0) {
          int rc = trace_planes(s0 + i, slot + i);
          if (rc) return rc;
        }
    if (timing_on) {
      float t03 = 0, t12 = 0;
      (void)hipEventElapsedTime(&t03, c->ev[0], c->ev[3]);
      (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
      c->ms_total += t03;
      c->ms_factor += t12;
    }
    if (mode == MODE_GRAD) {  // the dominant single kernel: the lauum launch of each group
      for (int g = 0; g <= gpc_ctx::MAXG; ++g)
        if (lauum_n[g] != 0) {
          float t = 0;
          if (timing_on) (void)hipEventElapsedTime(&t, c->ev_l0[g], c->ev_l1[g]);  // (else the stream was not waited for)
          if (t > c->ms_lauum) {
            c->ms_lauum = t;
            // (negative count: the timed launch is the dataflow graph -- factorization, inverse and W^T W of every
            // sample: S N^3 algorithmic flops)
            c->flops_lauum = lauum_n[g] > 0 ? (double)lauum_n[g] * (double)b.N * b.N * b.N / 3.0
                                            : (double)(-lauum_n[g]) * (double)b.N * b.N * b.N;
          }
          lauum_n[g] = 0;
        }
    }
    return 0;
  }

  // Jitter escalation (gaussian_process.py:2413-2421): every sample of [s0, s0+cnt) whose
  // factorization failed goes again with sn2_mult *= 10, up to 10 tries -- ALL failed samples of
  // a level as ONE batch (a 1024-point design with many non-PD rows costs at most 9 extra device
  // pipelines, not 9 per failed sample).  The sub-batch is computed in the scratch matrices
  // wA/wW/wT (slot i = i-th failed sample); commit(s, i) runs after a level for every sample s
  // that succeeded in slot i, before the next level overwrites the slots.
  template <typename Commit>
  int retry_failed(int s0, int cnt, T* wA, T* wW, T* wT, Commit&& commit) {
    Batch& b = *B;
    const int N = b.N, cov_N = b.cd.cov_N, Pn = P();
    const int nsn = b.vec_noise ? N : 1;
    std::vector<int> fail;
    for (int s = s0; s < s0 + cnt; ++s)
      if (b.info[s] != 0) {
        fail.push_back(s);
        b.tries[s] = 1;
      }
    // The attempt that failed ran in fast mode (trsm as a product with an explicit inverse: not backward stable when
    // the factor is ill-conditioned, plan.h).  The first retry repeats the SAME multiplier in stable mode -- where
    // LAPACK, i.e. the reference, may well succeed -- and every later level (x 10 each, ten levels in all like
    // :2413-2421) runs in stable mode too.
    bool same_level = !(stable || c->stable);
    while (!fail.empty() && (same_level || b.tries[fail[0]] < 10)) {
      const int nf = (int)fail.size();
      Batch sb;
      sb.S = nf;
      sb.N = N;
      sb.D = b.D;
      sb.npad = b.npad;
      sb.cd = b.cd;
      sb.vec_noise = b.vec_noise;
      sb.y = b.y;
      std::vector<double> hc(b.hyp_cov ? (size_t)nf * cov_N : 0), mm((size_t)nf * (b.m_const ? 1 : N)), sn((size_t)nf * nsn), gdm, gds;
      const bool gm = mode == MODE_GRAD && mean_N > 0 && dm, gn = mode == MODE_GRAD && noise_N > 0 && b.vec_noise;
      sb.m_const = b.m_const;
      if (gm) gdm.resize((size_t)nf * N * mean_N);
      if (gn) gds.resize((size_t)nf * N * noise_N);
      for (int i = 0; i < nf; ++i) {
        const int s = fail[i];
        if (b.hyp_cov) std::copy_n(b.hyp_cov + (size_t)s * cov_N, cov_N, &hc[(size_t)i * cov_N]);
        if (b.m_const)
          mm[i] = b.m[s];
        else
          std::copy_n(b.m + (size_t)s * N, N, &mm[(size_t)i * N]);
        std::copy_n(b.sn2 + (size_t)s * nsn, nsn, &sn[(size_t)i * nsn]);
        if (gm) std::copy_n(dm + (size_t)s * N * mean_N, (size_t)N * mean_N, &gdm[(size_t)i * N * mean_N]);
        if (gn) std::copy_n(dsn2 + (size_t)s * N * noise_N, (size_t)N * noise_N, &gds[(size_t)i * N * noise_N]);
      }
      sb.hyp_cov = b.hyp_cov ? hc.data() : nullptr;
      sb.m = mm.data();
      sb.sn2 = sn.data();
      sb.init();
      for (int i = 0; i < nf; ++i) {
        const int s = fail[i];
        if (!same_level) {
          b.mult[s] *= 10.0;
          b.tries[s] += 1;
          b.apply_mult(s);
        }
        sb.mult[i] = b.mult[s];
        sb.apply_mult(i);
      }
      same_level = false;
      Pipe<T> q;
      q.c = c;
      q.B = &sb;
      q.mode = mode;
      q.stable = true;
      q.A = wA;
      q.W = wW;
      q.Tm = wT;
      q.sM = sM;
      q.dm = gm ? gdm.data() : dm;  // scalar-noise dsn2 / absent dm are not read by the device
      q.mean_N = mean_N;
      q.dsn2 = gn ? gds.data() : dsn2;
      q.noise_N = noise_N;
      if (kmode()) {
        q.dk_cb = dk_cb;
        q.dk_user = dk_user;
        for (int i = 0; i < nf; ++i) {
          q.Kptr.push_back(Kptr[fail[i]]);
          q.orig.push_back(orig.empty() ? fail[i] : orig[fail[i]]);
        }
      }
      q.logdet.assign(nf, 0.0);
      q.quad.assign(nf, 0.0);
      q.G.assign((size_t)nf * Pn, 0.0);
      q.mg.assign((size_t)nf * std::max(mean_N, 1), 0.0);
      q.ng.assign((size_t)nf * std::max(noise_N, 1), 0.0);
      int rc = q.run(0, nf, 0);
      if (rc) return rc;
      ++c->retry_runs;
      std::vector<int> still;
      for (int i = 0; i < nf; ++i) {
        const int s = fail[i];
        b.info[s] = sb.info[i];
        if (sb.info[i] != 0) {
          still.push_back(s);
          continue;
        }
        logdet[s] = q.logdet[i];
        quad[s] = q.quad[i];
        if (mode == MODE_GRAD) {
          std::copy_n(&q.G[(size_t)i * Pn], Pn, &G[(size_t)s * Pn]);
          if (mean_N > 0) std::copy_n(&q.mg[(size_t)i * mean_N], mean_N, &mg[(size_t)s * mean_N]);
          if (noise_N > 0) std::copy_n(&q.ng[(size_t)i * noise_N], noise_N, &ng[(size_t)s * noise_N]);
        }
        rc = commit(s, i);
        if (rc) return rc;
      }
      fail.swap(still);
    }
    return 0;
  }
};

size_t free_device_bytes() {
  // GPC_MEM_BUDGET_MB caps what the library considers free (tests use it to force the
  // sample-chunking paths)
  if (const char* e = getenv("GPC_MEM_BUDGET_MB")) return (size_t)atoll(e) << 20;
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) return (size_t)8 << 30;
  return f;
}

// caller-provided covariance matrices (K-mode): S matrices of N x N doubles, and the dK plane callback
struct KArgs {
  const double* K = nullptr;
  gpc_dk_plane_fn cb = nullptr;
  void* user = nullptr;
};
template <typename PipeT>
void set_kmode(PipeT& p, const KArgs* km, int S, int N) {
  if (!km) return;
  for (int s = 0; s < S; ++s) p.Kptr.push_back(km->K + (size_t)s * N * N);
  p.dk_cb = km->cb;
  p.dk_user = km->user;
}

template <typename T>
int nll_impl(gpc_ctx* c, Batch& b, int want_grad, const double* dm, int mean_N, const double* dsn2,
             int noise_N, double* nlz, double* dnlz, double* sn2_mult, int* L_chol, int* info,
             const KArgs* km = nullptr) {
  const int S = b.S, npad = b.npad, N = b.N;
  const size_t per = 3ull * npad * npad * sizeof(T);
  int chunk = S;
  const bool forced = getenv("GPC_MEM_BUDGET_MB") != nullptr;
  const size_t one = (size_t)npad * npad * sizeof(T);
  if (forced || (size_t)S * one > std::min(c->mA.bytes, std::min(c->mW.bytes, c->mT.bytes))) {
    // the workspace must grow: size the chunk to what is free (hipMemGetInfo is slow, so it is
    // only consulted here)
    c->pool_drain();
    size_t budget = free_device_bytes() + (forced ? 0 : c->mA.bytes + c->mW.bytes + c->mT.bytes);
    budget = (size_t)(budget * 0.8);
    chunk = (int)std::max<size_t>(1, std::min<size_t>(S, budget / per));
  }
  HIPCHK(c, c->mA.ensure((size_t)chunk * npad * npad * sizeof(T)));
  HIPCHK(c, c->mW.ensure((size_t)chunk * npad * npad * sizeof(T)));
  HIPCHK(c, c->mT.ensure((size_t)chunk * npad * npad * sizeof(T)));

  Pipe<T> p;
  p.c = c;
  p.B = &b;
  p.mode = want_grad ? MODE_GRAD : MODE_NLL;
  p.A = c->mA.as<T>();
  p.W = c->mW.as<T>();
  p.Tm = c->mT.as<T>();
  p.sM = (long long)npad * npad;
  p.dm = dm;
  p.mean_N = mean_N;
  p.dsn2 = dsn2;
  p.noise_N = noise_N;
  set_kmode(p, km, S, N);
  const int Pn = p.P();
  p.logdet.assign(S, 0.0);
  p.quad.assign(S, 0.0);
  p.G.assign((size_t)S * Pn, 0.0);
  p.mg.assign((size_t)S * std::max(mean_N, 1), 0.0);
  p.ng.assign((size_t)S * std::max(noise_N, 1), 0.0);

  c->ms_total = c->ms_factor = 0;
  c->ms_lauum = c->flops_lauum = 0;
  c->last_flops = 0;
  c->retry_runs = 0;
  for (int s0 = 0; s0 < S; s0 += chunk) {
    const int cnt = std::min(chunk, S - s0);
    int rc = p.run(s0, cnt, 0);
    if (rc) return rc;
    rc = p.retry_failed(s0, cnt, p.A, p.W, p.Tm, [](int, int) { return 0; });
    if (rc) return rc;
  }
  const int cov_N = b.cd.cov_N;
  const int hyp_N = cov_N + noise_N + mean_N;
  for (int s = 0; s < S; ++s) {
    const double sl = b.sl[s];
    nlz[s] = 0.5 * p.quad[s] / sl + p.logdet[s] + N * std::log(2 * M_PI * sl) / 2;
    sn2_mult[s] = b.mult[s];
    L_chol[s] = b.lchol[s];
    info[s] = b.info[s];
    if (want_grad) {
      double* g = dnlz + (size_t)s * hyp_N;
      for (int h = 0; h < cov_N; ++h) g[h] = p.G[(size_t)s * Pn + h] / 2;  // :2487-2488
      const double trQ = p.G[(size_t)s * Pn + cov_N];
      for (int i = 0; i < noise_N; ++i) {
        if (b.vec_noise)
          g[cov_N + i] = 0.5 * b.mult[s] * p.ng[(size_t)s * noise_N + i];  // :2500-2504
        else
          g[cov_N + i] = 0.5 * b.mult[s] * dsn2[(size_t)s * noise_N + i] * trQ;  // :2491-2498
      }
      for (int i = 0; i < mean_N; ++i) g[cov_N + noise_N + i] = -p.mg[(size_t)s * mean_N + i];  // :2507-2508
    }
  }
  return 0;
}

int check_batch_args(gpc_ctx* c, int kernel_id, int degree, int dtype, int S) {
  if (!c) return -2;
  if (c->N <= 0) FAIL(c, "gpc_set_data has not been called");
  if (kernel_id != -1 && !valid_kernel(kernel_id, degree)) FAIL(c, "unknown covariance kernel / degree");
  if (dtype != GPC_F64 && dtype != GPC_F32) FAIL(c, "dtype must be GPC_F64 or GPC_F32");
  if (S <= 0) FAIL(c, "S must be positive");
  if (c->N > gpc_max_n(dtype))
    FAIL(c, "N is beyond what this device's memory holds for this dtype (three padded N x N slabs of one sample must fit "
            "in 80 % of it: gpc_max_n)");
  return 0;
}

void fill_batch(gpc_ctx* c, Batch& b, int kernel_id, int degree, int S, const double* hyp_cov,
                const double* m, const double* sn2, int vec, bool m_const = false) {
  b.m_const = m_const;
  b.S = S;
  b.N = c->N;
  b.D = c->D;
  b.npad = c->npad;
  b.cd.kind = kernel_id;
  b.cd.degree = degree;
  b.cd.D = c->D;
  b.cd.cov_N = kernel_id < 0 ? degree : cov_count_of(kernel_id, c->D);  // K-mode: `degree` carries cov_N
  b.vec_noise = vec != 0;
  b.hyp_cov = hyp_cov;
  b.m = m;
  b.sn2 = sn2;
  b.y = c->hy.data();
  b.mult0 = c->start_mult;
  b.init();
}

template <typename T>
int post_impl(gpc_ctx* c, Batch& b, gpc_post* po, double* sn2_mult, int* L_chol, int* info,
              const KArgs* km = nullptr) {
  const int S = b.S, npad = b.npad;
  const size_t msz = (size_t)npad * npad * sizeof(T);
  HostClock hc("post");
  po->A = c->pool_take(S * msz);
  po->W = c->pool_take(S * msz);
  po->alpha = c->pool_take((size_t)S * npad * sizeof(double));
  if (!po->A.p || !po->W.p) c->pool_drain();  // nothing reusable: give the memory back before growing
  HIPCHK(c, po->A.ensure(S * msz));
  HIPCHK(c, po->W.ensure(S * msz));
  HIPCHK(c, po->alpha.ensure((size_t)S * npad * sizeof(double)));
  hc.lap("alloc A,W");
  int chunk = S;
  if (getenv("GPC_MEM_BUDGET_MB") || (size_t)S * msz > c->mT.bytes) {
    c->pool_drain();
    size_t budget = (size_t)((free_device_bytes() + (getenv("GPC_MEM_BUDGET_MB") ? 0 : c->mT.bytes)) * 0.8);
    chunk = (int)std::max<size_t>(1, std::min<size_t>(S, budget / msz));
  }
  HIPCHK(c, c->mT.ensure((size_t)chunk * msz));
  hc.lap("alloc T");

  Pipe<T> p;
  p.c = c;
  p.B = &b;
  p.mode = MODE_POST;
  p.sM = (long long)npad * npad;
  set_kmode(p, km, S, b.N);
  p.logdet.assign(S, 0.0);
  p.quad.assign(S, 0.0);
  c->ms_total = c->ms_factor = 0;
  c->last_flops = 0;
  c->retry_runs = 0;
  for (int s0 = 0; s0 < S; s0 += chunk) {
    const int cnt = std::min(chunk, S - s0);
    // matrices of sample s live at po->A + s*sM; the pipe indexes from `slot`
    p.A = po->A.as<T>() + (size_t)s0 * p.sM;
    p.W = po->W.as<T>() + (size_t)s0 * p.sM;
    p.Tm = c->mT.as<T>();
    int rc = p.run(s0, cnt, 0);
    if (rc) return rc;
    auto save_alpha = [&](int s, int src_slot) -> int {
      HIPCHK(c, hipMemcpyAsync(po->alpha.as<double>() + (size_t)s * npad,
                               c->avec.as<double>() + (size_t)src_slot * npad, npad * sizeof(double),
                               hipMemcpyDeviceToDevice, c->st));
      return 0;
    };
    int nfail = 0;
    for (int i = 0; i < cnt; ++i) {
      if (b.info[s0 + i] == 0 && save_alpha(s0 + i, i)) return -1;
      nfail += b.info[s0 + i] != 0;
    }
    if (nfail) {  // retried in scratch matrices, successes copied into the posterior's slots
      HIPCHK(c, c->mA.ensure((size_t)nfail * msz));
      HIPCHK(c, c->mW.ensure((size_t)nfail * msz));
      rc = p.retry_failed(s0, cnt, c->mA.as<T>(), c->mW.as<T>(), c->mT.as<T>(), [&](int s, int i) -> int {
        HIPCHK(c, hipMemcpyAsync(po->A.as<T>() + (size_t)s * p.sM, c->mA.as<T>() + (size_t)i * p.sM, msz,
                                 hipMemcpyDeviceToDevice, c->st));
        HIPCHK(c, hipMemcpyAsync(po->W.as<T>() + (size_t)s * p.sM, c->mW.as<T>() + (size_t)i * p.sM, msz,
                                 hipMemcpyDeviceToDevice, c->st));
        return save_alpha(s, i);
      });
      if (rc) return rc;
    }
    // low-noise samples: Posterior.L = -(K + mult*Sigma)^-1  (gaussian_process.py:2441-2448)
    for (int s = s0; s < s0 + cnt; ++s)
      if (!b.lchol[s] && b.info[s] == 0) {
        Factor<T> F;
        F.st = c->st;
        F.batch = 1;
        F.npad = npad;
        F.A = po->A.as<T>() + (size_t)s * p.sM;
        F.W = po->W.as<T>() + (size_t)s * p.sM;
        F.Tm = c->mT.as<T>();
        F.sA = F.sW = F.sT = p.sM;
        F.lauum(F.Tm, p.sM);
        HIPCHK(c, F.err);
        dim3 g1(npad / 64, npad / 4, 1), blk(64, 4);
        hipLaunchKernelGGL((neg_sym_kernel<T>), g1, blk, 0, c->st, F.Tm, p.sM, npad, npad);
        HIPCHK(c, hipMemcpyAsync(F.A, F.Tm, msz, hipMemcpyDeviceToDevice, c->st));
        HIPCHK(c, hipGetLastError());
      }
    HIPCHK(c, hipStreamSynchronize(c->st));
  }
  hc.lap("device");
  po->dev_consts = false;
  po->sp = b.sp;
  po->mul = b.mul;
  po->dv = b.dv;
  po->mult = b.mult;
  po->lchol = b.lchol;
  po->info = b.info;
  po->sW.resize(S);
  for (int s = 0; s < S; ++s) {
    po->sW[s] = 1.0 / std::sqrt(b.smin[s] * b.mult[s]);  // :2517
    sn2_mult[s] = b.mult[s];
    L_chol[s] = b.lchol[s];
    info[s] = b.info[s];
  }
  return 0;
}

// Shared by predict / predict_full / quad: for every posterior sample s build a right-hand
// side matrix R_s (npad x mpad: cross covariances, or quadrature kernel means), then
//   lin[j*S+s]  = R_s[:, j] . alpha_s
//   quad[j*S+s] = |W_s R_s[:, j]|^2          (L_chol;  the caller divides by sl)
//               = R_s[:, j] . (L_s R_s[:, j]) (low noise, L = -inv)
//   full[s]     = Kss_s - (W R)^T (W R) / sl   or   Kss_s + R^T (L R)     (mode_full)
// mode: 0 = cross covariance of xa (M x D) with the training inputs; 1 = quadrature
// vectors z for Gaussian measures N(xa[j], diag(xb[j]^2)) (gaussian_process.py:1908-1921).
template <typename T>
int rhs_products(gpc_post* po, int mode, const double* xa, const double* xb, int M, bool want_quad,
                 double* lin, double* quad, double* full) {
  gpc_ctx* c = po->ctx;
  const int S = po->S, N = po->N, D = po->D, npad = po->npad;
  const int mpad = pad_tile(M);
  hipStream_t st = c->st;
  const long long sM = (long long)npad * npad;
  const long long sKs = (long long)npad * mpad;
  const long long sKss = (long long)mpad * mpad;
  const size_t per = (2ull * npad * mpad + (full ? (size_t)mpad * mpad : 0)) * sizeof(T);
  int chunk = S;
  if (getenv("GPC_MEM_BUDGET_MB") || (size_t)S * per > c->ks.bytes + c->vb.bytes + c->kss.bytes ||
      (size_t)S * sKs * sizeof(T) > std::min(c->ks.bytes, c->vb.bytes) ||
      (full && (size_t)S * sKss * sizeof(T) > c->kss.bytes)) {
    c->pool_drain();
    size_t budget = (size_t)((free_device_bytes() +
                              (getenv("GPC_MEM_BUDGET_MB") ? 0 : c->ks.bytes + c->vb.bytes + c->kss.bytes)) * 0.8);
    chunk = (int)std::max<size_t>(1, std::min<size_t>(S, budget / per));
  }
  HIPCHK(c, c->ks.ensure((size_t)chunk * sKs * sizeof(T)));
  HIPCHK(c, c->vb.ensure((size_t)chunk * sKs * sizeof(T)));
  if (full) HIPCHK(c, c->kss.ensure((size_t)chunk * sKss * sizeof(T)));
  HIPCHK(c, c->xss.ensure(((size_t)chunk * mpad * D + 2 * (size_t)M * D) * 8));
  HIPCHK(c, c->xs.ensure((size_t)chunk * npad * D * 8));
  HIPCHK(c, c->spb.ensure((size_t)chunk * SP_STRIDE * 8));
  HIPCHK(c, c->mulb.ensure((size_t)chunk * D * 8));
  HIPCHK(c, c->divb.ensure((size_t)chunk * D * 8));
  HIPCHK(c, c->pout.ensure((size_t)chunk * mpad * 2 * 8));
  double* d_xa = c->xss.as<double>() + (size_t)chunk * mpad * D;
  double* d_xb = d_xa + (size_t)M * D;
  c->pin.begin();
  c->pin.begin_gather();
  // (round 6, as for the evaluations: below N_pad = 2048 -- the posterior of a small training set queried in a loop by an
  // acquisition function -- timing events are recorded on request only and the call returns on the polled word of its
  // download launch)
  const bool timing_on = c->small_timing != 0 || npad >= 2048;
  if (mode != 2) {
    HIPCHK(c, c->pin.up(d_xa, xa, (size_t)M * D * 8, st));
    if (xb) HIPCHK(c, c->pin.up(d_xb, xb, (size_t)M * D * 8, st));
  }
  // landing buffers of the per-chunk results (pinned when available)
  // (one block for both: the device keeps [mu | v] contiguous too, so a call's results come back in ONE copy)
  std::vector<double> hmu_v;
  double* hmu = static_cast<double*>(c->pin.alloc(2 * (size_t)chunk * mpad * 8));
  if (!hmu) {
    hmu_v.resize(2 * (size_t)chunk * mpad);
    hmu = hmu_v.data();
  }
  double* hv = hmu + (size_t)chunk * mpad;
  double* hfull = nullptr;
  if (full && (size_t)M * M * 8 >= PinBuf::kMin && (size_t)M * M * 8 <= PinBuf::kMax)
    hfull = static_cast<double*>(c->pin.alloc((size_t)M * M * 8));

  // all samples in one chunk (the usual case): the posterior's constants are resident, see gpc_post
  const bool resident = chunk == S && mode != 2;
  if (resident && !po->dev_consts) {
    HIPCHK(c, po->dsp.ensure_private((size_t)S * SP_STRIDE * 8));
    HIPCHK(c, po->dmul.ensure_private((size_t)S * D * 8));
    HIPCHK(c, po->ddv.ensure_private((size_t)S * D * 8));
    HIPCHK(c, po->dxs.ensure_private((size_t)S * npad * D * 8));
    HIPCHK(c, hipMemcpyAsync(po->dsp.p, po->sp.data(), (size_t)S * SP_STRIDE * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(po->dmul.p, po->mul.data(), (size_t)S * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(po->ddv.p, po->dv.data(), (size_t)S * D * 8, hipMemcpyHostToDevice, st));
    const long long tot = (long long)npad * D;
    hipLaunchKernelGGL(scale_x_kernel, dim3((unsigned)((tot + 255) / 256), S), dim3(256), 0, st, c->dX.as<double>(), N,
                       npad, D, po->dmul.as<double>(), po->ddv.as<double>(), po->dxs.as<double>());
    HIPCHK(c, hipGetLastError());
    po->dev_consts = true;
  }
  const double* spb = resident ? po->dsp.as<double>() : c->spb.as<double>();
  const double* mulb = resident ? po->dmul.as<double>() : c->mulb.as<double>();
  const double* divb = resident ? po->ddv.as<double>() : c->divb.as<double>();
  const double* xsb = resident ? po->dxs.as<double>() : c->xs.as<double>();

  // gpc_last_timing after a predict / predict_full / quad call: device time of the call (first upload to last result
  // ready) and of its N^2 M products V = W Ks (the launches of gemm.h), summed over the chunks
  c->ms_total = c->ms_factor = 0;
  for (int s0 = 0; s0 < S; s0 += chunk) {
    const int cnt = std::min(chunk, S - s0);
    if (timing_on) HIPCHK(c, hipEventRecord(c->ev[0], st));
    if (!resident) {
      HIPCHK(c, hipMemcpyAsync(c->spb.p, &po->sp[(size_t)s0 * SP_STRIDE], (size_t)cnt * SP_STRIDE * 8,
                               hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(c->mulb.p, &po->mul[(size_t)s0 * D], (size_t)cnt * D * 8, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(c->divb.p, &po->dv[(size_t)s0 * D], (size_t)cnt * D * 8, hipMemcpyHostToDevice, st));
    }
    T* Ks = c->ks.as<T>();
    T* V = c->vb.as<T>();
    bool fused_mu = false;
    if (mode == 0) {
      long long tot = (long long)npad * D;
      if (!resident)
        hipLaunchKernelGGL(scale_x_kernel, dim3((unsigned)((tot + 255) / 256), cnt), dim3(256), 0, st,
                           c->dX.as<double>(), N, npad, D, mulb, divb, c->xs.as<double>());
      tot = (long long)mpad * D;
      hipLaunchKernelGGL(scale_x_kernel, dim3((unsigned)((tot + 255) / 256), cnt), dim3(256), 0, st,
                         (const double*)d_xa, M, mpad, D, mulb, divb, c->xss.as<double>());
      // cross covariances in 64 x 64 tiles with the mean product fused in (covfun.h: cross_tile_kernel)
      HIPCHK(c, c->dbg2.ensure((size_t)cnt * (npad / CT) * mpad * 8));
      GPC_COV_DISPATCH(cross_tile_kernel, T, po->cd, dim3(mpad / CT, npad / CT, cnt), dim3(256), 0, st, po->cd, xsb,
                       (const double*)c->xss.as<double>(), spb, (const double*)(po->alpha.as<double>() + (size_t)s0 * npad),
                       npad, N, npad, M, mpad, Ks, sKs, c->dbg2.as<double>());
      fused_mu = true;
      if (full)
        hipLaunchKernelGGL((cross_kernel<T>), dim3(mpad / 64, mpad / 4, cnt), dim3(64, 4), 0, st, po->cd,
                           c->xss.as<double>(), c->xss.as<double>(), spb, M, mpad, M, mpad, c->kss.as<T>(), sKss);
    } else if (mode == 2) {
      // caller-provided cross covariances Ks_s (N x M doubles, xa) and, with `full`, K**_s (M x M, xb)
      HIPCHK(c, c->dbg1.ensure((size_t)std::max(N, M) * M * sizeof(double)));
      for (int i = 0; i < cnt; ++i) {
        HIPCHK(c, hipMemcpyAsync(c->dbg1.p, xa + (size_t)(s0 + i) * N * M, (size_t)N * M * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((pad_rect_kernel<T>), dim3(mpad / 64, npad / 4), dim3(64, 4), 0, st, (const double*)c->dbg1.p, N,
                           M, npad, mpad, Ks + (size_t)i * sKs);
        if (full) {
          HIPCHK(c, hipMemcpyAsync(c->dbg1.p, xb + (size_t)(s0 + i) * M * M, (size_t)M * M * 8, hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL((pad_rect_kernel<T>), dim3(mpad / 64, mpad / 4), dim3(64, 4), 0, st, (const double*)c->dbg1.p,
                             M, M, mpad, mpad, c->kss.as<T>() + (size_t)i * sKss);
        }
      }
    } else {
      hipLaunchKernelGGL((quad_z_kernel<T>), dim3(mpad / 64, npad / 4, cnt), dim3(64, 4), 0, st,
                         c->dX.as<double>(), (const double*)d_xa, (const double*)d_xb, mulb, divb, spb, N, npad, M, mpad,
                         D, Ks, sKs);
    }
    double* d_mu = c->pout.as<double>();
    double* d_v = d_mu + (size_t)chunk * mpad;
    if (fused_mu)
      hipLaunchKernelGGL(colpart_reduce_kernel, dim3((mpad + 255) / 256, cnt), dim3(256), 0, st,
                         (const double*)c->dbg2.as<double>(), npad / CT, mpad, d_mu);
    else
      hipLaunchKernelGGL((colsum_vec_kernel<T>), dim3(mpad / 64, cnt), dim3(256), 0, st, (const T*)Ks, sKs, mpad,
                         po->alpha.as<double>() + (size_t)s0 * npad, npad, npad, mpad, d_mu);
    HIPCHK(c, hipGetLastError());
    // runs of equal L_chol share launches
    int a = 0;
    if (timing_on) {
      HIPCHK(c, hipEventRecord(c->ev[1], st));
      HIPCHK(c, hipEventRecord(c->ev[2], st));
    }
    while ((want_quad || full) && a < cnt) {
      int e = a;
      while (e < cnt && po->lchol[s0 + e] == po->lchol[s0 + a]) ++e;
      const int len = e - a;
      const bool lch = po->lchol[s0 + a] != 0;
      GemmArgs g;
      g.B = Ks + (size_t)a * sKs;
      g.C = V + (size_t)a * sKs;
      g.sB = g.sC = sKs;
      g.sA = sM;
      g.lda = npad;
      g.ldb = g.ldc = mpad;
      g.M = npad;
      g.N = mpad;
      g.K = npad;
      g.alpha = 1.0;
      g.beta = 0;
      g.klo = KLO_ZERO;
      g.lower_only = 0;
      g.A = (lch ? po->W.as<T>() : po->A.as<T>()) + (size_t)(s0 + a) * sM;  // V = W R | G = L R
      g.khi = lch ? KHI_ROW : KHI_FULL;
      // (problems whose product is a handful of 128-tiles keep the 64-tile product + column-sum pass: latency regime.
      // Decided by the problem size only, never by the number of samples: the two forms add in different orders, and a
      // sample of a batch must carry the bits of its single evaluation)
      const long long tiles128 = (long long)(npad / TILE) * (mpad / TILE);
      if (lch && !full && tiles128 >= 64) {
        // the variance needs the column sums of squares of V only: the product's epilogue forms them per tile row and V
        // is never written (gemm.h: EPI = 1); a small reduction over the tile rows follows
        const int tm = npad / TILE;
        HIPCHK(c, c->dbg3.ensure((size_t)cnt * tm * mpad * 8));  // (the whole chunk: no reallocation between runs)
        int* qctr = c->tile_ctr.as<int>() + (size_t)gpc_ctx::MAXG * gpc_ctx::CTR_PER_GROUP;
        HIPCHK(c, hipMemsetAsync(qctr, 0, CTR_STRIDE * sizeof(int), st));
        if (timing_on) HIPCHK(c, hipEventRecord(c->ev[1], st));
        g.colsq = c->dbg3.as<double>();
        HIPCHK(c, launch_gemm_colsq<T>(st, g, len, qctr));
        if (e == cnt && timing_on) HIPCHK(c, hipEventRecord(c->ev[2], st));
        hipLaunchKernelGGL(colpart_reduce_kernel, dim3((mpad + 255) / 256, len), dim3(256), 0, st,
                           (const double*)c->dbg3.as<double>(), tm, mpad, d_v + (size_t)a * mpad);
        a = e;
        continue;
      }
      HIPCHK(c, launch_gemm<T>(st, g, false, true, len));
      if (e == cnt && timing_on) HIPCHK(c, hipEventRecord(c->ev[2], st));  // (several runs: the first launch to the last, with what lies between)
      const T* left = lch ? (const T*)(V + (size_t)a * sKs) : (const T*)(Ks + (size_t)a * sKs);
      hipLaunchKernelGGL((colsum_prod_kernel<T>), dim3(mpad / 64, len), dim3(256), 0, st, left, sKs,
                         (const T*)(V + (size_t)a * sKs), sKs, mpad, npad, mpad, d_v + (size_t)a * mpad);
      if (full) {
        // Kss -= V^T V / sl  (per sample: alpha differs)   |   Kss += R^T G
        for (int i = a; i < e; ++i) {
          GemmArgs f;
          f.A = lch ? (const void*)(V + (size_t)i * sKs) : (const void*)(Ks + (size_t)i * sKs);
          f.B = V + (size_t)i * sKs;
          f.C = c->kss.as<T>() + (size_t)i * sKss;
          f.sA = f.sB = f.sC = 0;
          f.lda = f.ldb = f.ldc = mpad;
          f.M = f.N = mpad;
          f.K = npad;
          f.alpha = lch ? -1.0 / po->sp[(size_t)(s0 + i) * SP_STRIDE + SP_SL] : 1.0;
          f.beta = 1;
          f.klo = KLO_ZERO;
          f.khi = KHI_FULL;
          f.lower_only = 0;
          HIPCHK(c, launch_gemm<T>(st, f, true, true, 1));
        }
      }
      a = e;
    }
    HIPCHK(c, hipGetLastError());
    const size_t out_bytes = (want_quad ? 2 : 1) * (size_t)chunk * mpad * 8;
    const bool poll = !timing_on && !full && c->small_poll && c->land_blk && cnt == chunk && out_bytes <= PinBuf::kGather / 8;
    if (poll) {  // the results through the gathered download launch, which carries the completion word
      HIPCHK(c, c->pin.gather(hmu, d_mu, want_quad ? out_bytes : (size_t)cnt * mpad * 8, st));
    } else if (want_quad && cnt == chunk)  // d_v = d_mu + chunk * mpad: one contiguous block
      HIPCHK(c, hipMemcpyAsync(hmu, d_mu, 2 * (size_t)chunk * mpad * 8, hipMemcpyDeviceToHost, st));
    else {
      HIPCHK(c, hipMemcpyAsync(hmu, d_mu, (size_t)cnt * mpad * 8, hipMemcpyDeviceToHost, st));
      if (want_quad) HIPCHK(c, hipMemcpyAsync(hv, d_v, (size_t)cnt * mpad * 8, hipMemcpyDeviceToHost, st));
    }
    if (full) {
      HIPCHK(c, c->dbg3.ensure((size_t)M * M * 8));
      for (int i = 0; i < cnt; ++i) {
        dim3 gn((M + 63) / 64, (M + 3) / 4), blk(64, 4);
        hipLaunchKernelGGL((extract_kernel<T>), gn, blk, 0, st, (const T*)(c->kss.as<T>() + (size_t)i * sKss),
                           mpad, M, 2, c->dbg3.as<double>());
        double* dst = full + (size_t)(s0 + i) * M * M;
        HIPCHK(c, hipMemcpyAsync(hfull ? hfull : dst, c->dbg3.p, (size_t)M * M * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (hfull) memcpy(dst, hfull, (size_t)M * M * 8);
      }
    }
    if (timing_on) HIPCHK(c, hipEventRecord(c->ev[3], st));
    bool seen = false;
    if (poll && !c->pin.down_plain) {
      unsigned long long* flag = reinterpret_cast<unsigned long long*>(c->land_blk + (gpc_ctx::LAND_BYTES / 8 - 1));
      const unsigned long long seq = ++c->land_seq;
      HIPCHK(c, c->pin.flush_down(st, flag, seq, nullptr));
      const volatile unsigned long long* fw = flag;
      const auto t_poll = std::chrono::steady_clock::now();
      for (int spin = 0; !seen; ++spin) {
        if (*fw == seq) {
          seen = true;
          break;
        }
        if ((spin & 63) == 63 &&
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_poll).count() > 2000.0)
          break;
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    } else if (poll) {
      HIPCHK(c, c->pin.flush_down(st));
    }
    if (seen) {
      ++c->small_polled;
    } else {
      HIPCHK(c, hipStreamSynchronize(st));
      if (poll) ++c->small_synced;
    }
    c->pin.finish();
    if (timing_on) {
      float t03 = 0, t12 = 0;
      (void)hipEventElapsedTime(&t03, c->ev[0], c->ev[3]);
      (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
      c->ms_total += t03;
      c->ms_factor += t12;
    }
    for (int i = 0; i < cnt; ++i) {
      const int s = s0 + i;
      for (int j = 0; j < M; ++j) {
        lin[(size_t)j * S + s] = hmu[(size_t)i * mpad + j];
        if (want_quad) quad[(size_t)j * S + s] = hv[(size_t)i * mpad + j];
      }
    }
  }
  return 0;
}

template <typename T>
int predict_impl(gpc_post* po, const double* xstar, int M, double* fmu, double* fs2) {
  int rc = rhs_products<T>(po, 0, xstar, nullptr, M, true, fmu, fs2, nullptr);
  if (rc) return rc;
  const int S = po->S;
  for (int s = 0; s < S; ++s) {
    const double sf2 = po->sp[(size_t)s * SP_STRIDE + SP_SF2];
    const double sl = po->sp[(size_t)s * SP_STRIDE + SP_SL];
    for (int j = 0; j < M; ++j) {
      const double q = fs2[(size_t)j * S + s];
      // L_chol: kss - sum(V*V), V = sW * (W Ks), sW^2 = 1/sl   (:1752-1760)
      // else  : kss + sum(Ks * (L Ks))                          (:1762-1764)
      fs2[(size_t)j * S + s] = po->lchol[s] ? sf2 - q / sl : sf2 + q;
    }
  }
  return 0;
}

}  // namespace

// test-hook helpers
namespace {
template <typename T>
int upload_as(gpc_ctx* c, DevBuf& buf, const double* src, size_t n) {
  HIPCHK(c, buf.ensure(n * sizeof(T)));
  if constexpr (sizeof(T) == 8) {
    HIPCHK(c, hipMemcpyAsync(buf.p, src, n * 8, hipMemcpyHostToDevice, c->st));
  } else {
    std::vector<float> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
    HIPCHK(c, hipMemcpyAsync(buf.p, tmp.data(), n * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
  }
  return 0;
}
template <typename T>
int download_as(gpc_ctx* c, const T* src, double* dst, size_t n) {
  if constexpr (sizeof(T) == 8) {
    HIPCHK(c, hipMemcpyAsync(dst, src, n * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
  } else {
    std::vector<float> tmp(n);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), src, n * 4, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    for (size_t i = 0; i < n; ++i) dst[i] = tmp[i];
  }
  return 0;
}

template <typename T>
int debug_gemm_impl(gpc_ctx* c, int M, int N, int K, int akm, int bkm, double alpha, int beta, int klo,
                    int khi, int lower, const double* A, const double* B, double* C) {
  if (upload_as<T>(c, c->dbg1, A, (size_t)M * K)) return -1;
  if (upload_as<T>(c, c->dbg2, B, (size_t)N * K)) return -1;
  if (upload_as<T>(c, c->dbg3, C, (size_t)M * N)) return -1;
  GemmArgs g;
  g.A = c->dbg1.p;
  g.B = c->dbg2.p;
  g.C = c->dbg3.p;
  g.sA = g.sB = g.sC = 0;
  g.lda = akm ? M : K;
  g.ldb = bkm ? N : K;
  g.ldc = N;
  g.M = M;
  g.N = N;
  g.K = K;
  g.alpha = alpha;
  g.beta = beta;
  g.klo = klo;
  g.khi = khi;
  g.lower_only = lower & 1;
  g.tiles_n = N / TILE;
  HIPCHK(c, launch_gemm<T>(c->st, g, akm != 0, bkm != 0, 1, (lower & 0x100) ? 64 : ((lower & 0x200) ? 128 : ((lower & 0x400) ? 12864 : 0))));  // (12864: experiments build)
  return download_as<T>(c, c->dbg3.as<T>(), C, (size_t)M * N);
}

template <typename T>
int debug_factor_impl(gpc_ctx* c, int n, const double* A, double* L, double* W, double* Ainv,
                      double* logdet, int* info) {
  const int npad = pad_tile(n);
  const size_t msz = (size_t)npad * npad;
  HIPCHK(c, c->mA.ensure(msz * sizeof(T)));
  HIPCHK(c, c->mW.ensure(msz * sizeof(T)));
  HIPCHK(c, c->mT.ensure(msz * sizeof(T)));
  HIPCHK(c, c->dbg1.ensure((size_t)n * n * 8));
  HIPCHK(c, c->scal.ensure(64));
  HIPCHK(c, hipMemcpyAsync(c->dbg1.p, A, (size_t)n * n * 8, hipMemcpyHostToDevice, c->st));
  HIPCHK(c, hipMemsetAsync(c->scal.p, 0, 64, c->st));
  dim3 blk(64, 4), gp(npad / 64, npad / 4);
  hipLaunchKernelGGL((pad_load_kernel<T>), gp, blk, 0, c->st, c->dbg1.as<double>(), n, npad, c->mA.as<T>());
  Factor<T> F;
  F.st = c->st;
  F.batch = 1;
  F.npad = npad;
  F.A = c->mA.as<T>();
  F.W = c->mW.as<T>();
  F.Tm = c->mT.as<T>();
  F.sA = F.sW = F.sT = (long long)msz;
  F.logdet = c->scal.as<double>();
  F.info = reinterpret_cast<int*>(c->scal.as<double>() + 1);
  F.nvalid = n;
  F.potrf_inv(0, npad, true, true);
  HIPCHK(c, F.err);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, c->dbg3.ensure((size_t)n * n * 8));
  dim3 gn((n + 63) / 64, (n + 3) / 4);
  if (L) {
    hipLaunchKernelGGL((extract_kernel<T>), gn, blk, 0, c->st, (const T*)F.A, npad, n, 0, c->dbg3.as<double>());
    HIPCHK(c, hipMemcpyAsync(L, c->dbg3.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
  }
  if (W) {
    hipLaunchKernelGGL((extract_kernel<T>), gn, blk, 0, c->st, (const T*)F.W, npad, n, 0, c->dbg3.as<double>());
    HIPCHK(c, hipMemcpyAsync(W, c->dbg3.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
  }
  if (Ainv) {
    F.lauum(F.Tm, F.sT);
    HIPCHK(c, F.err);
    hipLaunchKernelGGL((extract_kernel<T>), gn, blk, 0, c->st, (const T*)F.Tm, npad, n, 0, c->dbg3.as<double>());
    HIPCHK(c, hipMemcpyAsync(Ainv, c->dbg3.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
  }
  double h[2];
  HIPCHK(c, hipMemcpyAsync(h, c->scal.p, 16, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, hipStreamSynchronize(c->st));
  if (logdet) *logdet = h[0];
  if (info) memcpy(info, &h[1], sizeof(int));
  return 0;
}
}  // namespace

namespace {
// rank-one append (scalar noise; checked by the caller): high-noise samples get a new last row
// of the factor and of its inverse (:776-817), low-noise samples a rank-one update of -inv (:819-827)
// Ks_h / kss_h: the cross covariances and prior variances at the new point from the caller's own covariance object
// (K-mode posteriors); nullptr: built on the device from the posterior's kernel
template <typename T>
int append_impl(gpc_post* po, const double* m_star, const double* sn2_star, double y_new, int* ok,
                const double* Ks_h = nullptr, const double* kss_h = nullptr) {
  gpc_ctx* c = po->ctx;
  const int S = po->S, D = po->D, n = po->N;  // the new point is row n of the context's X
  hipStream_t st = c->st;
  if (c->N != n + 1) FAIL(c, "gpc_post_append: call gpc_set_data with the extended X, y first");
  // grow the padded storage by one tile when the new row does not fit
  if (n + 1 > po->npad) {
    const int np = po->npad, npn = np + TILE;
    DevBuf nA, nW, nal;
    HIPCHK(c, nA.ensure((size_t)S * npn * npn * sizeof(T)));
    HIPCHK(c, nW.ensure((size_t)S * npn * npn * sizeof(T)));
    HIPCHK(c, nal.ensure((size_t)S * npn * sizeof(double)));
    dim3 g(npn / 64, npn / 4, S), blk(64, 4);
    hipLaunchKernelGGL((grow_copy_kernel<T>), g, blk, 0, st, (const T*)po->A.as<T>(), np, nA.as<T>(), npn);
    hipLaunchKernelGGL((grow_copy_kernel<T>), g, blk, 0, st, (const T*)po->W.as<T>(), np, nW.as<T>(), npn);
    HIPCHK(c, hipMemsetAsync(nal.p, 0, (size_t)S * npn * sizeof(double), st));
    HIPCHK(c, hipMemcpy2DAsync(nal.p, (size_t)npn * 8, po->alpha.p, (size_t)np * 8, (size_t)np * 8, S,
                               hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->pool_give(po->A);
    c->pool_give(po->W);
    c->pool_give(po->alpha);
    po->A = nA;
    po->W = nW;
    po->alpha = nal;
    po->npad = npn;
  }
  const int npad = po->npad;
  const long long sM = (long long)npad * npad;
  const size_t vb = (size_t)npad * 8;
  HIPCHK(c, c->xs.ensure((size_t)S * npad * D * 8));
  HIPCHK(c, c->spb.ensure((size_t)S * SP_STRIDE * 8));
  HIPCHK(c, c->mulb.ensure((size_t)S * D * 8));
  HIPCHK(c, c->divb.ensure((size_t)S * D * 8));
  HIPCHK(c, c->rvec.ensure(S * vb));
  HIPCHK(c, c->zvec.ensure(S * vb));
  HIPCHK(c, c->avec.ensure(S * vb));
  HIPCHK(c, c->tpart.ensure((size_t)S * (npad / TRC + 1) * vb));
  HIPCHK(c, c->scal.ensure((size_t)S * 10 * 8));
  HIPCHK(c, hipMemcpyAsync(c->spb.p, po->sp.data(), (size_t)S * SP_STRIDE * 8, hipMemcpyHostToDevice, st));
  if (!Ks_h) {
    HIPCHK(c, hipMemcpyAsync(c->mulb.p, po->mul.data(), (size_t)S * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->divb.p, po->dv.data(), (size_t)S * D * 8, hipMemcpyHostToDevice, st));
    const long long tot = (long long)npad * D;
    hipLaunchKernelGGL(scale_x_kernel, dim3((unsigned)((tot + 255) / 256), S), dim3(256), 0, st,
                       c->dX.as<double>(), n + 1, npad, D, c->mulb.as<double>(), c->divb.as<double>(),
                       c->xs.as<double>());
  }
  double* ks = c->rvec.as<double>();  // Ks
  double* lv = c->zvec.as<double>();  // l = W Ks
  double* au = c->avec.as<double>();  // W^T l
  double* d_ll = c->scal.as<double>();
  double* d_ka = d_ll + S;
  if (Ks_h) {  // rows of n values, zero beyond (the padding rows of W are identity rows)
    HIPCHK(c, hipMemsetAsync(ks, 0, S * vb, st));
    HIPCHK(c, hipMemcpy2DAsync(ks, vb, Ks_h, (size_t)n * 8, (size_t)n * 8, S, hipMemcpyHostToDevice, st));
  } else {
    hipLaunchKernelGGL(cross_vec_kernel, dim3((npad + 255) / 256, S), dim3(256), 0, st, po->cd,
                       c->xs.as<double>(), c->spb.as<double>(), n, npad, ks);
  }
  hipLaunchKernelGGL((trmv_kernel<T>), dim3(npad / 4, S), dim3(256), 0, st, (const T*)po->W.as<T>(), sM, npad,
                     (const double*)ks, npad, lv, 0);
  hipLaunchKernelGGL(dot_kernel, dim3(1, S), dim3(256), 0, st, (const double*)lv, (const double*)lv, n, npad, d_ll);
  hipLaunchKernelGGL(dot_kernel, dim3(1, S), dim3(256), 0, st, (const double*)ks,
                     (const double*)po->alpha.as<double>(), n, npad, d_ka);
  // the padding rows of W are identity rows: l[i >= n] = Ks[i] = 0, harmless in W^T l
  double* tpart = c->tpart.as<double>();
  hipLaunchKernelGGL((trmv_t_part_kernel<T>), dim3((npad + 64 * MM<T>::VEC - 1) / (64 * MM<T>::VEC), npad / TRC, S), dim3(256), 0, st,
                     (const T*)po->W.as<T>(), sM, npad, (const double*)lv, npad, tpart, (double*)nullptr);
  hipLaunchKernelGGL(trmv_t_sum_kernel, dim3(npad / 128, S), dim3(128), 0, st, (const double*)tpart, npad,
                     (const double*)nullptr, 0, 0, au);
  HIPCHK(c, hipGetLastError());
  // low-noise samples (Posterior.L = -(K + Sigma)^-1, a FULL symmetric matrix; :819-827):
  //   au = -L Ks,  Ks.au (-> predictive variance),  per sample (their W is not used)
  double* au_low = c->dvec.as<double>();
  double* d_kau = d_ka + S;
  HIPCHK(c, c->dvec.ensure(S * vb));
  au_low = c->dvec.as<double>();
  bool any_low = false;
  for (int s = 0; s < S; ++s) {
    if (po->lchol[s]) continue;
    any_low = true;
    HIPCHK(c, hipMemsetAsync(au_low + (size_t)s * npad, 0, vb, st));
    hipLaunchKernelGGL((gemv_sub_kernel<T>), dim3(npad / 4, 1), dim3(256), 0, st,
                       (const T*)(po->A.as<T>() + (size_t)s * sM), 0ll, npad, (const double*)(ks + (size_t)s * npad),
                       au_low + (size_t)s * npad, npad, 0, 0, npad);
  }
  if (any_low)
    hipLaunchKernelGGL(dot_kernel, dim3(1, S), dim3(256), 0, st, (const double*)ks, (const double*)au_low, n, npad,
                       d_kau);
  HIPCHK(c, hipGetLastError());
  std::vector<double> ll(S), ka(S), kau(S, 0.0), coef((size_t)S * 5);
  HIPCHK(c, hipMemcpyAsync(ll.data(), d_ll, S * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(ka.data(), d_ka, S * 8, hipMemcpyDeviceToHost, st));
  if (any_low) HIPCHK(c, hipMemcpyAsync(kau.data(), d_kau, S * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  for (int s = 0; s < S; ++s) {
    const double sf2 = kss_h ? kss_h[s] : po->sp[(size_t)s * SP_STRIDE + SP_SF2];  // k(x_new, x_new)
    const double sl = po->sp[(size_t)s * SP_STRIDE + SP_SL];  // = sn2 * sn2_mult of the fitted noise (L_chol)
    const double sn2_eff = sn2_star[s] * po->mult[s];
    const double mu_star = m_star[s] + ka[s];  // predictive mean at the new point
    double* cf = &coef[(size_t)s * 5];
    ok[s] = 0;
    if (po->info[s] != 0 || ((c->append_fail_mask >> (s & 31)) & 1u)) continue;
    if (po->lchol[s]) {
      // gaussian_process.py:784-788 (K = kss = sf2)
      const double sqrt_arg = sn2_eff * sn2_eff + sf2 * sn2_eff - ll[s];
      if (!(sqrt_arg > 0.0 && std::abs(sn2_eff - sl) <= 1e-12 * sl)) continue;
      const double dl = std::sqrt(sqrt_arg) / sn2_eff;   // new diagonal entry of the factor (:814)
      const double v_star = sf2 - ll[s] / sl + sn2_eff;  // predictive variance incl. noise (:756)
      cf[0] = 1.0 / sn2_eff;                             // Lo[n][:n] = l / sn2_eff  (:811)
      cf[1] = dl;
      cf[2] = -1.0 / (dl * sn2_eff);                     // W[n][:n] = -(l/sn2_eff)^T W / dl
      cf[3] = (mu_star - y_new) / v_star;                // alpha update weight (:842)
      cf[4] = 1.0 / sn2_eff;                             // alpha_update = W^T l / sn2_eff (:800-808)
    } else {
      // s2 = kss + Ks.(L Ks) = sf2 - Ks.au, clamped at 0 (:1762-1770), plus the noise (:1779)
      const double v_star = std::max(sf2 - kau[s], 0.0) + sn2_eff;
      cf[0] = 1.0 / v_star;                // v = -au / v_star (:821); new L = [[L + v au^T, -v], [-v^T, -1/v_star]]
      cf[3] = (mu_star - y_new) / v_star;  // alpha += cf3 * au, alpha[n] = -cf3 (:840-844)
    }
    ok[s] = 1;
  }
  double* d_coef = c->scal.as<double>() + 3 * S;
  HIPCHK(c, hipMemcpyAsync(d_coef, coef.data(), coef.size() * 8, hipMemcpyHostToDevice, st));
  // Samples that cannot be appended keep their old rows: the caller recomputes them on the grown
  // storage (gpc_post_recompute) -- the reference's per-posterior fallback (:789-798, :866-869).
  for (int s = 0; s < S; ++s) {
    if (!ok[s]) continue;
    if (po->lchol[s])
      hipLaunchKernelGGL((append_row_kernel<T>), dim3((n + 256) / 256, 1), dim3(256), 0, st,
                         po->A.as<T>() + (size_t)s * sM, po->W.as<T>() + (size_t)s * sM, sM, npad, n,
                         (const double*)(lv + (size_t)s * npad), (const double*)(au + (size_t)s * npad), npad,
                         (const double*)(d_coef + (size_t)s * 5), po->alpha.as<double>() + (size_t)s * npad);
    else
      hipLaunchKernelGGL((append_low_kernel<T>), dim3((n + 64) / 64, (n + 4) / 4), dim3(64, 4), 0, st,
                         po->A.as<T>() + (size_t)s * sM, npad, n, (const double*)(au_low + (size_t)s * npad),
                         (const double*)(d_coef + (size_t)s * 5), po->alpha.as<double>() + (size_t)s * npad);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(st));
  po->N = n + 1;
  po->dev_consts = false;
  return 0;
}
}  // namespace


// =====================================================================================
// C ABI
// =====================================================================================
namespace {
template <typename T>
__global__ void tile_hash_kernel(const T* __restrict__ M, int npad, unsigned long long* __restrict__ out) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  unsigned long long h = 0;
  for (int e = threadIdx.x; e < TILE * TILE; e += 256) {
    const T v = M[(size_t)(ti * TILE + e / TILE) * npad + tj * TILE + e % TILE];
    unsigned long long b = 0;
    memcpy(&b, &v, sizeof(T));
    h += b * (unsigned long long)(2 * e + 1);
  }
  __shared__ unsigned long long sh[256];
  sh[threadIdx.x] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[ti * gridDim.x + tj] = sh[0];
}
}  // namespace

extern "C" {

int gpc_create(int device, gpc_ctx** out) {
  if (!out) return -2;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_err = std::string("no HIP device: ") + hipGetErrorString(e);
    return -1;
  }
  if (device < 0 || device >= count) {
    g_create_err = "device index out of range";
    return -2;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    g_create_err = hipGetErrorString(e);
    return -1;
  }
  gpc_ctx* c = new gpc_ctx();
  c->device = device;
  hipDeviceProp_t prop{};
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s arch=%s CUs=%d clock=%dMHz mem=%.1fGB lds/block=%zuKB", prop.name,
             prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000,
             prop.totalGlobalMem / 1073741824.0, prop.sharedMemPerBlock / 1024);
    c->devinfo = buf;
  }
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // lo = least, hi = greatest priority
  if (hipStreamCreateWithPriority(&c->st, hipStreamNonBlocking, prio_hi) != hipSuccess) {
    g_create_err = "hipStreamCreate failed";
    delete c;
    return -1;
  }
  for (auto& ev : c->ev)
    if (hipEventCreate(&ev) != hipSuccess) {
      g_create_err = "hipEventCreate failed";
      delete c;
      return -1;
    }
  bool ok = hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming) == hipSuccess;
  for (int g = 0; g <= gpc_ctx::MAXG && ok; ++g)
    ok = hipEventCreate(&c->ev_l0[g]) == hipSuccess && hipEventCreate(&c->ev_l1[g]) == hipSuccess;
  for (int g = 0; g < gpc_ctx::MAXG && ok; ++g)
    ok = hipStreamCreateWithPriority(&c->gst[g], hipStreamNonBlocking, prio_hi) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_done[g], hipEventDisableTiming) == hipSuccess;
  for (int g = 0; g <= gpc_ctx::MAXG && ok; ++g) {
    ok = hipStreamCreateWithPriority(&c->sst[g], hipStreamNonBlocking, prio_lo) == hipSuccess;
    for (int i = 0; i < gpc_ctx::NDEV && ok; ++i)
      ok = hipEventCreateWithFlags(&c->dev_ev[g][i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_bfork[g], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_btail[g], hipEventDisableTiming) == hipSuccess;
  }
  if (const char* e = getenv("GPC_DEFER_MIN")) c->defer_min = atoi(e);
  if (const char* e = getenv("GPC_DEFER_RESERVE")) c->defer_reserve = atoi(e);
  if (!ok) {
    g_create_err = "creating the sample-group streams failed";
    delete c;
    return -1;
  }
  if (const char* e = getenv("GPC_PERSIST_SPARE")) gpc::g_persist_spare = atoi(e);  // < 0: no persistent launches
  gpc::g_block_slots = 2 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
  if (c->tile_ctr.ensure((gpc_ctx::MAXG + 1) * gpc_ctx::CTR_PER_GROUP * sizeof(int)) != hipSuccess) {
    g_create_err = "allocating the tile counters failed";
    delete c;
    return -1;
  }
  {  // the landing block of one-leaf evaluations: coherent (fine-grained) host memory, so that a word the device writes
     // in the middle of a kernel is seen by a polling host; without it those calls wait on the stream as before
    void* lp = nullptr;
    if (hipHostMalloc(&lp, gpc_ctx::LAND_BYTES, hipHostMallocCoherent) == hipSuccess) {
      memset(lp, 0, gpc_ctx::LAND_BYTES);
      c->land_blk = static_cast<double*>(lp);
    } else {
      (void)hipGetLastError();
    }
  }
  if (const char* e = getenv("GPC_SMALL_POLL")) c->small_poll = atoi(e) != 0;
  if (probe_cu_map(c, prop.multiProcessorCount) != 0 || build_reserve_table(c) != 0) {
    g_create_err = "probing the CU map failed: " + c->err;
    delete c;
    return -1;
  }
  if (const char* e = getenv("GPC_LEAF")) gpc::g_leaf_version = atoi(e) == 3 ? 3 : 5;
  if (const char* e = getenv("GPC_SMALL_PATH")) c->small_path = atoi(e) != 0;
  if (const char* e = getenv("GPC_GRAPH_MAX_NPAD")) c->graph_max_npad = atoi(e);
#ifdef GPC_EXPERIMENTS
  if (const char* e = getenv("GPC_RL_PANEL")) c->rl_panel = atoi(e) <= 0 ? 0 : std::max(TILE, (atoi(e) / TILE) * TILE);
  if (const char* e = getenv("GPC_RECT_MIN")) gpc::g_rect_min_blocks = atoi(e);
#endif
  if (const char* e = getenv("GPC_NLL_BLOCK")) c->nll_block = atoi(e) < 0 ? -1 : (atoi(e) == 0 ? 0 : std::max(TILE, (atoi(e) / TILE) * TILE));
  if (const char* e = getenv("GPC_GROUPS")) c->groups = std::max(1, std::min((int)gpc_ctx::MAXG, atoi(e)));
  if (const char* e = getenv("GPC_SMALL_BLOCKS")) gpc::g_small_launch_blocks = atoi(e);
  if (const char* e = getenv("GPC_DUAL")) gpc::g_dual_launch = atoi(e) != 0;
  if (const char* e = getenv("GPC_GEMM_FLAGS")) gpc::g_gemm_flags = atoi(e);
  if (const char* e = getenv("GPC_XCD_AFFINE")) gpc::g_gemm_flags = atoi(e) ? (gpc::g_gemm_flags | 8) : (gpc::g_gemm_flags & ~8);
  *out = c;
  return 0;
}

void gpc_destroy(gpc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->st);
  DevBuf* bufs[] = {&c->dX,   &c->dY,  &c->mA,    &c->mW,  &c->mT,   &c->xs,   &c->spb,  &c->mulb, &c->divb,
                    &c->dvec, &c->rvec,  &c->zvec, &c->avec, &c->scal, &c->parts, &c->gout, &c->diagq,
                    &c->dmb,  &c->dsn2b, &c->mg,  &c->ng,   &c->ks,   &c->vb,   &c->xss,  &c->pout, &c->kss,
                    &c->dbg1, &c->dbg2,  &c->dbg3, &c->tpart, &c->tile_ctr, &c->rsv_tbl};
  for (auto& g : c->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  for (DevBuf* b : bufs) b->release();
  c->pool_drain();
  c->pin.release();
  if (c->land_blk) (void)hipHostFree(c->land_blk);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (c->ev_up) (void)hipEventDestroy(c->ev_up);
  for (int g = 0; g <= gpc_ctx::MAXG; ++g) {
    if (c->ev_l0[g]) (void)hipEventDestroy(c->ev_l0[g]);
    if (c->ev_l1[g]) (void)hipEventDestroy(c->ev_l1[g]);
  }
  for (int g = 0; g <= gpc_ctx::MAXG; ++g) {
    for (int i = 0; i < gpc_ctx::NDEV; ++i)
      if (c->dev_ev[g][i]) (void)hipEventDestroy(c->dev_ev[g][i]);
    if (c->ev_bfork[g]) (void)hipEventDestroy(c->ev_bfork[g]);
    if (c->ev_btail[g]) (void)hipEventDestroy(c->ev_btail[g]);
    if (c->sst[g]) (void)hipStreamDestroy(c->sst[g]);
  }
  for (int g = 0; g < gpc_ctx::MAXG; ++g) {
    if (c->ev_done[g]) (void)hipEventDestroy(c->ev_done[g]);
    if (c->gst[g]) (void)hipStreamDestroy(c->gst[g]);
  }
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

const char* gpc_last_error(const gpc_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

const char* gpc_device_info(gpc_ctx* c) { return c ? c->devinfo.c_str() : ""; }

int gpc_cov_count(int kernel_id, int D) { return cov_count_of(kernel_id, D); }

int gpc_max_n(int dtype) {
  // A memory-budget answer (round 6; rounds 1-5: 16384 / 23168, the reach of one 32-bit byte offset over a k-major
  // operand panel, which gemm.h no longer has): the largest N, a multiple of 128, whose three padded slabs A, W and
  // scratch (plan.h) of ONE sample fit in 80 % of the current device's memory -- what nll_impl / post_impl budget a chunk
  // with.  Without a device: the MI355X's 288 GB.
  const long long w = dtype == GPC_F32 ? 4 : 8;
  size_t fr = 0, total = 0;
  if (hipMemGetInfo(&fr, &total) != hipSuccess || total == 0) {
    (void)hipGetLastError();
    total = (size_t)288 << 30;
  }
  const double budget = 0.8 * (double)total / (3.0 * (double)w);
  long long n = (long long)std::floor(std::sqrt(budget) / TILE) * TILE;
  // (tile and element counts are 32-bit in the kernels' index arithmetic up to (npad / 64)^2 tiles: far beyond any memory)
  return (int)std::max<long long>(TILE, std::min<long long>(n, 1 << 20));
}

int gpc_set_data(gpc_ctx* c, const double* X, const double* y, int N, int D) {
  if (!c) return -2;
  if (!X || !y || N <= 0 || D <= 0) FAIL(c, "gpc_set_data: X, y must be non-null and N, D positive");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->dX.ensure((size_t)N * D * sizeof(double)));
  c->pin.begin();
  HIPCHK(c, c->pin.up(c->dX.p, X, (size_t)N * D * sizeof(double), c->st));
  HIPCHK(c, c->dY.ensure((size_t)N * sizeof(double)));
  HIPCHK(c, hipMemcpyAsync(c->dY.p, y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPCHK(c, hipStreamSynchronize(c->st));
  c->hy.assign(y, y + N);
  c->N = N;
  c->D = D;
  c->npad = pad_tile(N);
  return 0;
}

int gpc_kernel(gpc_ctx* c, int kernel_id, int degree, const double* hyp_cov, const double* X, int N, int D,
               const double* Xstar, int M, int diag, double* K, double* dK) {
  if (!c) return -2;
  if (!valid_kernel(kernel_id, degree)) FAIL(c, "unknown covariance kernel / degree");
  if (!hyp_cov || !X || !K || N <= 0 || D <= 0) FAIL(c, "gpc_kernel: bad arguments");
  if (Xstar && dK) FAIL(c, "X_star should be None when compute_grad is True.");
  HIPCHK(c, hipSetDevice(c->device));
  CovDesc cd{kernel_id, degree, D, cov_count_of(kernel_id, D)};
  std::vector<double> mul(D), dv(D);
  double sf2, rqa;
  scaling_of(kernel_id, degree, D, hyp_cov, mul.data(), dv.data(), &sf2, &rqa);
  if (diag) {  // zero distance: covariance_functions.py:162-163
    for (int i = 0; i < N; ++i) K[i] = sf2;
    return 0;
  }
  const int Mc = Xstar ? M : N;
  if (Mc <= 0) FAIL(c, "gpc_kernel: M must be positive with Xstar");
  hipStream_t st = c->st;
  // dbg1: raw X | raw X* | mul | div ; dbg2: scaled Xa | scaled Xb ; dbg3: K | dK
  const size_t nraw = (size_t)N * D + (size_t)Mc * D + 2 * (size_t)D;
  HIPCHK(c, c->dbg1.ensure(nraw * 8));
  HIPCHK(c, c->dbg2.ensure(((size_t)N * D + (size_t)Mc * D) * 8));
  const size_t nK = (size_t)N * Mc, ndK = dK ? nK * cd.cov_N : 0;
  HIPCHK(c, c->dbg3.ensure((nK + ndK) * 8));
  double* d_xa = c->dbg1.as<double>();
  double* d_xb = d_xa + (size_t)N * D;
  double* d_mul = d_xb + (size_t)Mc * D;
  double* d_div = d_mul + D;
  c->pin.begin();
  HIPCHK(c, c->pin.up(d_xa, X, (size_t)N * D * 8, st));
  HIPCHK(c, c->pin.up(d_xb, Xstar ? Xstar : X, (size_t)Mc * D * 8, st));
  HIPCHK(c, hipMemcpyAsync(d_mul, mul.data(), D * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_div, dv.data(), D * 8, hipMemcpyHostToDevice, st));
  double* s_xa = c->dbg2.as<double>();
  double* s_xb = s_xa + (size_t)N * D;
  hipLaunchKernelGGL(scale_x_kernel, dim3((unsigned)(((long long)N * D + 255) / 256), 1), dim3(256), 0, st,
                     (const double*)d_xa, N, N, D, (const double*)d_mul, (const double*)d_div, s_xa);
  hipLaunchKernelGGL(scale_x_kernel, dim3((unsigned)(((long long)Mc * D + 255) / 256), 1), dim3(256), 0, st,
                     (const double*)d_xb, Mc, Mc, D, (const double*)d_mul, (const double*)d_div, s_xb);
  double* d_K = c->dbg3.as<double>();
  double* d_dK = dK ? d_K + nK : nullptr;
  hipLaunchKernelGGL(full_cov_kernel, dim3((Mc + 63) / 64, (N + 3) / 4), dim3(64, 4), 0, st, cd,
                     (const double*)s_xa, (const double*)s_xb, sf2, rqa, N, Mc, d_K, d_dK);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, c->pin.down(K, d_K, nK * 8, st));
  if (dK) HIPCHK(c, c->pin.down(dK, d_dK, ndK * 8, st));
  HIPCHK(c, hipStreamSynchronize(st));
  c->pin.finish();
  return 0;
}

int gpc_nll_batch(gpc_ctx* c, int kernel_id, int degree, int dtype, int S, const double* hyp_cov,
                  const double* m, const double* sn2, int sn2_is_vector, int want_grad, const double* dm,
                  int mean_N, const double* dsn2, int noise_N, double* nlz, double* dnlz, double* sn2_mult,
                  int* L_chol, int* info) {
  int rc = check_batch_args(c, kernel_id, degree, dtype, S);
  if (rc) return rc;
  if (!hyp_cov || !m || !sn2 || !nlz || !sn2_mult || !L_chol || !info) FAIL(c, "gpc_nll_batch: null argument");
  if (want_grad && (!dnlz || (mean_N > 0 && !dm) || (noise_N > 0 && !dsn2)))
    FAIL(c, "gpc_nll_batch: gradient requested without dnlz/dm/dsn2");
  HIPCHK(c, hipSetDevice(c->device));
  HostClock hc("nll");
  Batch b;
  fill_batch(c, b, kernel_id, degree, S, hyp_cov, m, sn2, sn2_is_vector);
  hc.lap("fill_batch");
  if (dtype == GPC_F64)
    return nll_impl<double>(c, b, want_grad, dm, mean_N, dsn2, noise_N, nlz, dnlz, sn2_mult, L_chol, info);
  return nll_impl<float>(c, b, want_grad, dm, mean_N, dsn2, noise_N, nlz, dnlz, sn2_mult, L_chol, info);
}

int gpc_nll_batch_cm(gpc_ctx* c, int kernel_id, int degree, int dtype, int S, const double* hyp_cov,
                     const double* m0, int mean_N, const double* sn2, int sn2_is_vector, int want_grad,
                     const double* dsn2, int noise_N, double* nlz, double* dnlz, double* sn2_mult, int* L_chol,
                     int* info) {
  int rc = check_batch_args(c, kernel_id, degree, dtype, S);
  if (rc) return rc;
  if (!hyp_cov || !sn2 || !nlz || !sn2_mult || !L_chol || !info) FAIL(c, "gpc_nll_batch_cm: null argument");
  if (mean_N < 0 || mean_N > 1 || (mean_N == 1 && !m0)) FAIL(c, "gpc_nll_batch_cm: mean_N must be 0 (zero mean) or 1 (m0 given)");
  if (want_grad && (!dnlz || (noise_N > 0 && !dsn2))) FAIL(c, "gpc_nll_batch_cm: gradient requested without dnlz/dsn2");
  HIPCHK(c, hipSetDevice(c->device));
  HostClock hc("nll");
  std::vector<double> zero;
  if (!m0) {
    zero.assign(S, 0.0);
    m0 = zero.data();
  }
  Batch b;
  fill_batch(c, b, kernel_id, degree, S, hyp_cov, m0, sn2, sn2_is_vector, true);
  hc.lap("fill_batch");
  // dm = nullptr with mean_N = 1: the constant mean's derivative (ones), summed on the device
  if (dtype == GPC_F64)
    return nll_impl<double>(c, b, want_grad, nullptr, mean_N, dsn2, noise_N, nlz, dnlz, sn2_mult, L_chol, info);
  return nll_impl<float>(c, b, want_grad, nullptr, mean_N, dsn2, noise_N, nlz, dnlz, sn2_mult, L_chol, info);
}

int gpc_posterior_batch(gpc_ctx* c, int kernel_id, int degree, int dtype, int S, const double* hyp_cov,
                        const double* m, const double* sn2, int sn2_is_vector, gpc_post** post,
                        double* sn2_mult, int* L_chol, int* info) {
  int rc = check_batch_args(c, kernel_id, degree, dtype, S);
  if (rc) return rc;
  if (!hyp_cov || !m || !sn2 || !post || !sn2_mult || !L_chol || !info) FAIL(c, "gpc_posterior_batch: null argument");
  HIPCHK(c, hipSetDevice(c->device));
  Batch b;
  fill_batch(c, b, kernel_id, degree, S, hyp_cov, m, sn2, sn2_is_vector);
  gpc_post* po = new gpc_post();
  po->ctx = c;
  po->dtype = dtype;
  po->S = S;
  po->N = c->N;
  po->D = c->D;
  po->npad = c->npad;
  po->cd = b.cd;
  rc = (dtype == GPC_F64) ? post_impl<double>(c, b, po, sn2_mult, L_chol, info)
                          : post_impl<float>(c, b, po, sn2_mult, L_chol, info);
  if (rc) {
    c->pool_give(po->A);
    c->pool_give(po->W);
    c->pool_give(po->alpha);
    delete po;
    return rc;
  }
  *post = po;
  return 0;
}

int gpc_nll_batch_K(gpc_ctx* c, int dtype, int S, int cov_N, const double* K, gpc_dk_plane_fn dk_plane, void* user,
                    const double* m, const double* sn2, int sn2_is_vector, int want_grad, const double* dm,
                    int mean_N, const double* dsn2, int noise_N, double* nlz, double* dnlz, double* sn2_mult,
                    int* L_chol, int* info) {
  int rc = check_batch_args(c, -1, 0, dtype, S);
  if (rc) return rc;
  if (!K || !m || !sn2 || !nlz || !sn2_mult || !L_chol || !info || cov_N < 0) FAIL(c, "gpc_nll_batch_K: bad argument");
  if (want_grad && (!dnlz || (cov_N > 0 && !dk_plane) || (mean_N > 0 && !dm) || (noise_N > 0 && !dsn2)))
    FAIL(c, "gpc_nll_batch_K: gradient requested without dnlz / dK callback / dm / dsn2");
  HIPCHK(c, hipSetDevice(c->device));
  Batch b;
  fill_batch(c, b, -1, cov_N, S, nullptr, m, sn2, sn2_is_vector);
  KArgs km{K, dk_plane, user};
  if (dtype == GPC_F64)
    return nll_impl<double>(c, b, want_grad, dm, mean_N, dsn2, noise_N, nlz, dnlz, sn2_mult, L_chol, info, &km);
  return nll_impl<float>(c, b, want_grad, dm, mean_N, dsn2, noise_N, nlz, dnlz, sn2_mult, L_chol, info, &km);
}

int gpc_posterior_batch_K(gpc_ctx* c, int dtype, int S, const double* K, const double* m, const double* sn2,
                          int sn2_is_vector, gpc_post** post, double* sn2_mult, int* L_chol, int* info) {
  int rc = check_batch_args(c, -1, 0, dtype, S);
  if (rc) return rc;
  if (!K || !m || !sn2 || !post || !sn2_mult || !L_chol || !info) FAIL(c, "gpc_posterior_batch_K: null argument");
  HIPCHK(c, hipSetDevice(c->device));
  Batch b;
  fill_batch(c, b, -1, 0, S, nullptr, m, sn2, sn2_is_vector);
  gpc_post* po = new gpc_post();
  po->ctx = c;
  po->dtype = dtype;
  po->S = S;
  po->N = c->N;
  po->D = c->D;
  po->npad = c->npad;
  po->cd = b.cd;
  KArgs km{K, nullptr, nullptr};
  rc = (dtype == GPC_F64) ? post_impl<double>(c, b, po, sn2_mult, L_chol, info, &km)
                          : post_impl<float>(c, b, po, sn2_mult, L_chol, info, &km);
  if (rc) {
    c->pool_give(po->A);
    c->pool_give(po->W);
    c->pool_give(po->alpha);
    delete po;
    return rc;
  }
  *post = po;
  return 0;
}

int gpc_predict_K(gpc_post* po, int M, const double* Ks, const double* Kss, double* fmu, double* fq, double* cov) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (!Ks || !fmu || M <= 0 || (!fq && !cov) || (cov && !Kss)) FAIL(c, "gpc_predict_K: bad arguments");
  for (int s = 0; s < po->S; ++s)
    if (po->info[s] != 0) FAIL(c, "gpc_predict_K: posterior contains a failed factorization");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = po->dtype == GPC_F64 ? rhs_products<double>(po, 2, Ks, Kss, M, fq != nullptr, fmu, fq, cov)
                                : rhs_products<float>(po, 2, Ks, Kss, M, fq != nullptr, fmu, fq, cov);
  if (rc || !fq) return rc;
  for (int s = 0; s < po->S; ++s) {  // the term the caller adds to kss (:1752-1764)
    const double sl = po->sp[(size_t)s * SP_STRIDE + SP_SL];
    for (int j = 0; j < M; ++j) {
      double& q = fq[(size_t)j * po->S + s];
      q = po->lchol[s] ? -q / sl : q;
    }
  }
  return 0;
}

int gpc_post_fetch(gpc_post* po, int s, double* alpha, double* sW, double* L) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (s < 0 || s >= po->S) FAIL(c, "gpc_post_fetch: sample index out of range");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = po->N, npad = po->npad;
  c->pin.begin();
  if (alpha) HIPCHK(c, c->pin.down(alpha, po->alpha.as<double>() + (size_t)s * npad, N * sizeof(double), c->st));
  if (sW)
    for (int i = 0; i < N; ++i) sW[i] = po->sW[s];
  if (L) {
    HIPCHK(c, c->dbg3.ensure((size_t)N * N * 8));
    dim3 grid((N + 63) / 64, (N + 3) / 4), blk(64, 4);
    const int mode = po->lchol[s] ? 0 : 2;  // low-noise: A already holds the full -inv
    if (po->dtype == GPC_F64)
      hipLaunchKernelGGL((extract_kernel<double>), grid, blk, 0, c->st,
                         (const double*)(po->A.as<double>() + (size_t)s * npad * npad), npad, N, mode,
                         c->dbg3.as<double>());
    else
      hipLaunchKernelGGL((extract_kernel<float>), grid, blk, 0, c->st,
                         (const float*)(po->A.as<float>() + (size_t)s * npad * npad), npad, N, mode,
                         c->dbg3.as<double>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->pin.down(L, c->dbg3.p, (size_t)N * N * 8, c->st));
  }
  HIPCHK(c, hipStreamSynchronize(c->st));
  c->pin.finish();
  return 0;
}

int gpc_post_free(gpc_post* po) {
  if (!po) return 0;
  (void)hipSetDevice(po->ctx->device);
  (void)hipStreamSynchronize(po->ctx->st);
  po->ctx->pool_give(po->A);
  po->ctx->pool_give(po->W);
  po->ctx->pool_give(po->alpha);
  po->dsp.release();
  po->dmul.release();
  po->ddv.release();
  po->dxs.release();
  delete po;
  return 0;
}

int gpc_post_append(gpc_post* po, const double* m_star, const double* sn2_star, double y_new, int* ok) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (!m_star || !sn2_star || !ok) FAIL(c, "gpc_post_append: null argument");
  if (po->cd.kind < 0) FAIL(c, "gpc_post_append: not available for posteriors built from caller-provided K");
  HIPCHK(c, hipSetDevice(c->device));
  return po->dtype == GPC_F64 ? append_impl<double>(po, m_star, sn2_star, y_new, ok)
                              : append_impl<float>(po, m_star, sn2_star, y_new, ok);
}

int gpc_post_append_K(gpc_post* po, const double* Ks, const double* kss, const double* m_star,
                      const double* sn2_star, double y_new, int* ok) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (!Ks || !kss || !m_star || !sn2_star || !ok) FAIL(c, "gpc_post_append_K: null argument");
  if (po->cd.kind >= 0) FAIL(c, "gpc_post_append_K: this posterior was built from a device kernel; use gpc_post_append");
  HIPCHK(c, hipSetDevice(c->device));
  return po->dtype == GPC_F64 ? append_impl<double>(po, m_star, sn2_star, y_new, ok, Ks, kss)
                              : append_impl<float>(po, m_star, sn2_star, y_new, ok, Ks, kss);
}

namespace {
// full recompute of the listed samples of a resident posterior set, in place (device kernel: hyp_cov; K-mode: K)
int recompute_impl(gpc_post* po, int cnt, const int* idx, const double* hyp_cov, const double* K, const double* m,
                   const double* sn2, int sn2_is_vector, double* sn2_mult, int* L_chol, int* info) {
  gpc_ctx* c = po->ctx;
  if (c->N != po->N || c->D != po->D) FAIL(c, "gpc_post_recompute: the context's data do not match the posterior");
  for (int i = 0; i < cnt; ++i)
    if (idx[i] < 0 || idx[i] >= po->S) FAIL(c, "gpc_post_recompute: sample index out of range");
  HIPCHK(c, hipSetDevice(c->device));
  Batch b;
  if (K)
    fill_batch(c, b, -1, 0, cnt, nullptr, m, sn2, sn2_is_vector);
  else
    fill_batch(c, b, po->cd.kind, po->cd.degree, cnt, hyp_cov, m, sn2, sn2_is_vector);
  gpc_post tmp;
  tmp.ctx = c;
  tmp.dtype = po->dtype;
  tmp.S = cnt;
  tmp.N = c->N;
  tmp.D = c->D;
  tmp.npad = c->npad;
  tmp.cd = b.cd;
  if (tmp.npad != po->npad) FAIL(c, "gpc_post_recompute: padded size mismatch");
  KArgs km{K, nullptr, nullptr};
  int rc = (po->dtype == GPC_F64) ? post_impl<double>(c, b, &tmp, sn2_mult, L_chol, info, K ? &km : nullptr)
                                  : post_impl<float>(c, b, &tmp, sn2_mult, L_chol, info, K ? &km : nullptr);
  if (rc == 0) {
    const size_t w = po->dtype == GPC_F64 ? 8 : 4, msz = (size_t)po->npad * po->npad * w;
    for (int i = 0; i < cnt && rc == 0; ++i) {
      const int s = idx[i];
      hipError_t e = hipMemcpyAsync((char*)po->A.p + (size_t)s * msz, (char*)tmp.A.p + (size_t)i * msz, msz,
                                    hipMemcpyDeviceToDevice, c->st);
      if (e == hipSuccess)
        e = hipMemcpyAsync((char*)po->W.p + (size_t)s * msz, (char*)tmp.W.p + (size_t)i * msz, msz,
                           hipMemcpyDeviceToDevice, c->st);
      if (e == hipSuccess)
        e = hipMemcpyAsync(po->alpha.as<double>() + (size_t)s * po->npad, tmp.alpha.as<double>() + (size_t)i * po->npad,
                           (size_t)po->npad * 8, hipMemcpyDeviceToDevice, c->st);
      if (e != hipSuccess) {
        c->err = std::string("gpc_post_recompute copy: ") + hipGetErrorString(e);
        rc = -1;
        break;
      }
      po->dev_consts = false;
      std::copy_n(&tmp.sp[(size_t)i * SP_STRIDE], SP_STRIDE, &po->sp[(size_t)s * SP_STRIDE]);
      std::copy_n(&tmp.mul[(size_t)i * po->D], po->D, &po->mul[(size_t)s * po->D]);
      std::copy_n(&tmp.dv[(size_t)i * po->D], po->D, &po->dv[(size_t)s * po->D]);
      po->mult[s] = tmp.mult[i];
      po->lchol[s] = tmp.lchol[i];
      po->info[s] = tmp.info[i];
      po->sW[s] = tmp.sW[i];
    }
    (void)hipStreamSynchronize(c->st);
  }
  c->pool_give(tmp.A);
  c->pool_give(tmp.W);
  c->pool_give(tmp.alpha);
  return rc;
}
}  // namespace

int gpc_post_recompute(gpc_post* po, int cnt, const int* idx, const double* hyp_cov, const double* m,
                       const double* sn2, int sn2_is_vector, double* sn2_mult, int* L_chol, int* info) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (cnt <= 0 || !idx || !hyp_cov || !m || !sn2 || !sn2_mult || !L_chol || !info)
    FAIL(c, "gpc_post_recompute: bad arguments");
  if (po->cd.kind < 0) FAIL(c, "gpc_post_recompute: this posterior was built from caller-provided K; use gpc_post_recompute_K");
  return recompute_impl(po, cnt, idx, hyp_cov, nullptr, m, sn2, sn2_is_vector, sn2_mult, L_chol, info);
}

int gpc_post_recompute_K(gpc_post* po, int cnt, const int* idx, const double* K, const double* m, const double* sn2,
                         int sn2_is_vector, double* sn2_mult, int* L_chol, int* info) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (cnt <= 0 || !idx || !K || !m || !sn2 || !sn2_mult || !L_chol || !info)
    FAIL(c, "gpc_post_recompute_K: bad arguments");
  if (po->cd.kind >= 0) FAIL(c, "gpc_post_recompute_K: this posterior was built from a device kernel; use gpc_post_recompute");
  return recompute_impl(po, cnt, idx, nullptr, K, m, sn2, sn2_is_vector, sn2_mult, L_chol, info);
}

int gpc_predict(gpc_post* po, const double* xstar, int M, double* fmu, double* fs2) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (!xstar || !fmu || !fs2 || M <= 0) FAIL(c, "gpc_predict: bad arguments");
  if (po->cd.kind < 0) FAIL(c, "gpc_predict: this posterior was built from caller-provided K; use gpc_predict_K");
  for (int s = 0; s < po->S; ++s)
    if (po->info[s] != 0) FAIL(c, "gpc_predict: posterior contains a failed factorization");
  HIPCHK(c, hipSetDevice(c->device));
  return po->dtype == GPC_F64 ? predict_impl<double>(po, xstar, M, fmu, fs2)
                              : predict_impl<float>(po, xstar, M, fmu, fs2);
}

int gpc_predict_full(gpc_post* po, const double* xstar, int M, double* fmu, double* cov) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (!xstar || !fmu || !cov || M <= 0) FAIL(c, "gpc_predict_full: bad arguments");
  if (po->cd.kind < 0) FAIL(c, "gpc_predict_full: this posterior was built from caller-provided K; use gpc_predict_K");
  for (int s = 0; s < po->S; ++s)
    if (po->info[s] != 0) FAIL(c, "gpc_predict_full: posterior contains a failed factorization");
  HIPCHK(c, hipSetDevice(c->device));
  return po->dtype == GPC_F64 ? rhs_products<double>(po, 0, xstar, nullptr, M, false, fmu, nullptr, cov)
                              : rhs_products<float>(po, 0, xstar, nullptr, M, false, fmu, nullptr, cov);
}

int gpc_quad(gpc_post* po, const double* mu, const double* sigma, int M, int compute_var, double* zalpha,
             double* zKz) {
  if (!po) return -2;
  gpc_ctx* c = po->ctx;
  if (!mu || !sigma || !zalpha || M <= 0 || (compute_var && !zKz)) FAIL(c, "gpc_quad: bad arguments");
  if (po->cd.kind != K_SE && po->cd.kind != K_SE_ISO)
    FAIL(c, "Bayesian quadrature only supports the squared exponential kernel.");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = po->dtype == GPC_F64
               ? rhs_products<double>(po, 1, mu, sigma, M, compute_var != 0, zalpha, zKz, nullptr)
               : rhs_products<float>(po, 1, mu, sigma, M, compute_var != 0, zalpha, zKz, nullptr);
  if (rc || !compute_var) return rc;
  // z (K + sn2_eff I)^-1 z^T: |W z|^2 / sl (L_chol) or -(z . L z) (L = -inv)   (:1946-1962)
  for (int s = 0; s < po->S; ++s) {
    const double sl = po->sp[(size_t)s * SP_STRIDE + SP_SL];
    for (int j = 0; j < M; ++j) {
      double& q = zKz[(size_t)j * po->S + s];
      q = po->lchol[s] ? q / sl : -q;
    }
  }
  return 0;
}

int gpc_last_timing(gpc_ctx* c, double* ms_total, double* ms_factor) {
  if (!c) return -2;
  if (ms_total) *ms_total = c->ms_total;
  if (ms_factor) *ms_factor = c->ms_factor;
  return 0;
}

namespace {
#ifdef GPC_EXPERIMENTS
// options of the schedules that exist in the experiments build only (see the include of dag.h)
int* experiment_option(gpc_ctx* c, const std::string& n) {
  if (n == "rect_min") return &gpc::g_rect_min_blocks;  // launches of at least this many 128-tiles (x samples) below the 128-tile threshold run as 128 x 64 tiles (0: off)
  if (n == "rect_mode") return &gpc::g_rect_mode;       // what "rect_min" selects: 0 = 128 x 64 tiles of four waves, 1 = 128 x 128 tiles of eight waves
  if (n == "indep") return &c->indep;                   // independent pipelines for batches of 2 .. indep_max samples at npad >= 2048 (0: lock-step)
  if (n == "indep_max") return &c->indep_max;
  if (n == "indep_min_tiles") return &c->indep_min_tiles;
  if (n == "rl_ahead_max") return &c->rl_ahead_max;     // look-ahead of the right-looking plan only up to this batch work S (npad/4096)^3
  if (n == "rl_panel") return &c->rl_panel;             // NLL-only: right-looking panels of this many rows with look-ahead (0: off)
  if (n == "dag") return &c->dag;
  if (n == "dag_small_tiles") return &c->dag_small_tiles;
  if (n == "dag_lauum") return &c->dag_lauum;
  if (n == "dag_leaf_blocks") return &c->dag_leaf_blocks;
  if (n == "dag_aborts") return &c->dag_aborts;         // (tests: forget earlier aborts -- three of them switch the graph off for the context)
  if (n == "dag_runs") return &c->dag_runs;
  if (n == "dag_urgent_cus") return &c->dag_urgent_cus;
  if (n == "dag_gate") return &c->dag_gate;
  if (n == "dag_gate_pct") return &c->dag_gate_pct;
  if (n == "dag_crit_pct") return &c->dag_crit_pct;
  if (n == "dag_timeout_ms") return &c->dag_timeout_ms;
  return nullptr;
}
int clamp_experiment_option(const std::string& n, int value) {
  if (n == "indep_max") return std::max(2, std::min((int)gpc_ctx::MAXG, value));
  if (n == "indep_min_tiles") return std::max(1, value);
  if (n == "rect_mode") return value != 0;
  if (n == "rl_panel") return value <= 0 ? 0 : std::max(TILE, (value / TILE) * TILE);
  if (n == "dag_urgent_cus") return std::max(0, std::min(15, value));
  if (n == "dag_gate_pct" || n == "dag_crit_pct") return std::max(0, std::min(100, value));
  if (n == "dag_timeout_ms") return std::max(1, value);  // (a wait of no length would abort every graph at once: ADVICE r5)
  return value;
}
#endif
}  // namespace

int gpc_set_option(gpc_ctx* c, const char* name, int value) {
  if (!c || !name) return -2;
  const std::string n(name);
  if (n == "groups")
    c->groups = std::max(1, std::min((int)gpc_ctx::MAXG, value));
  else if (n == "small_blocks")
    gpc::g_small_launch_blocks = value;
  else if (n == "dual_launch")  // syrk + inverse product of a node in one launch (default 1)
    gpc::g_dual_launch = value != 0;
  else if (n == "leaf")  // 5: pipelined leaf (default), 3: barrier-per-phase leaf (A/B and bit-identity tests)
    gpc::g_leaf_version = value == 3 ? 3 : 5;
  else if (n == "defer_min")  // deferred inverse products: node size from which U runs on the side stream (0 off, -1 auto)
    c->defer_min = value;
  else if (n == "defer_reserve") {  // CUs per XCD a deferred launch stays off (2, 4, 8, 12; >= 32: all of them, a test hook)
    c->defer_reserve = value;
    if (build_reserve_table(c)) return -1;
  } else if (n == "leaf_fault")  // test hook: the pipelined leaf runs with a missing wave, its hand-offs time out
    gpc::g_leaf_fault = value != 0;
  else if (n == "nll_block")  // NLL-only: largest diagonal block with an inverse (multiple of 128; 0: left children inverted)
    c->nll_block = value < 0 ? -1 : (value == 0 ? 0 : std::max(TILE, (value / TILE) * TILE));
  else if (n == "solves_beside_lauum")  // 0: the triangular mat-vecs after the W^T W launch (round-2 order)
    c->solves_beside_lauum = value != 0;
  else if (n == "stable")  // every factorization in stable mode (refined panel solves, plan.h), not only the jitter retries
    c->stable = value != 0;
  else if (n == "small_path")  // 0: problems of one leaf take the general pipeline too (A/B and cross-checks)
    c->small_path = value != 0;
  else if (n == "check_queues")  // debug: verify the tile queues of persistent launches after every pipeline
    c->check_queues = value != 0;
  else if (n == "small_poll")  // 0: one-leaf evaluations wait on the stream instead of polling their landing block
    c->small_poll = value != 0;
  else if (n == "small_timing")  // 1: one-leaf evaluations record their timing events (gpc_last_timing is 0 for them otherwise)
    c->small_timing = value != 0;
  else if (n == "start_mult_log10")  // test hook: first jitter multiplier 10^value
    c->start_mult = std::pow(10.0, std::max(0, std::min(9, value)));
  else if (n == "append_fail_mask")  // test hook: samples whose rank-one append is declared unstable
    c->append_fail_mask = (unsigned)value;
  else {
#ifdef GPC_EXPERIMENTS
    if (int* slot = experiment_option(c, n))
      *slot = clamp_experiment_option(n, value);
    else
#endif
      FAIL(c, "gpc_set_option: unknown option");
  }
  ++g_alloc_epoch;  // cached launch graphs captured the old launch shapes
  return 0;
}

int gpc_get_option(gpc_ctx* c, const char* name, int* value) {
  if (!c || !name || !value) return -2;
  const std::string n(name);
  if (n == "groups") *value = c->groups;
  else if (n == "small_blocks") *value = gpc::g_small_launch_blocks;
  else if (n == "dual_launch") *value = gpc::g_dual_launch ? 1 : 0;
  else if (n == "leaf") *value = gpc::g_leaf_version;
  else if (n == "defer_min") *value = c->defer_min;
  else if (n == "defer_reserve") *value = c->defer_reserve;
  else if (n == "nll_block") *value = c->nll_block;
  else if (n == "solves_beside_lauum") *value = c->solves_beside_lauum;
  else if (n == "stable") *value = c->stable;
  else if (n == "small_path") *value = c->small_path;
  else if (n == "check_queues") *value = c->check_queues;
  else if (n == "small_poll") *value = c->small_poll;
  else if (n == "small_timing") *value = c->small_timing;
  else if (n == "small_polled") *value = (int)(c->small_polled & 0x7fffffff);  // one-leaf calls completed by the polled word ...
  else if (n == "small_synced") *value = (int)(c->small_synced & 0x7fffffff);  // ... and by a stream synchronisation
  else if (n == "experiments") {  // 1: this library is the experiments build (tests/ and tools/ ask before they use its options)
#ifdef GPC_EXPERIMENTS
    *value = 1;
#else
    *value = 0;
#endif
  } else {
#ifdef GPC_EXPERIMENTS
    if (int* slot = experiment_option(c, n))
      *value = *slot;
    else
#endif
      FAIL(c, "gpc_get_option: unknown option");
  }
  return 0;
}

int gpc_last_lauum_timing(gpc_ctx* c, double* ms, double* flops) {
  if (!c) return -2;
  if (ms) *ms = c->ms_lauum;
  if (flops) *flops = c->flops_lauum;
  return 0;
}

int gpc_mfma_peak(gpc_ctx* c, int dtype, double* tflops, double* cycles_per_mfma, double* clock_ghz) {
  if (!c || !tflops) return -2;
  HIPCHK(c, hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
  const int iters = 2048;
  HIPCHK(c, c->dbg1.ensure((size_t)prop.multiProcessorCount * 2 * 256 * 8));
  HIPCHK(c, c->dbg2.ensure(64));
  long long* d_clk = c->dbg2.as<long long>();
  double best = 0, best_cyc = 0, best_ghz = 0;
  if (dtype >= 2) {  // 2: f64 VALU FMA, 3: f32 VALU FMA (16 independent chains, 1 or 2 waves/SIMD)
    for (int wps : {1, 2}) {
      const int blocks = prop.multiProcessorCount * wps;
      auto launch = [&]() {
        if (dtype == 2)
          hipLaunchKernelGGL((valu_peak_kernel<double>), dim3(blocks), dim3(256), 0, c->st, c->dbg1.as<double>(), iters, d_clk);
        else
          hipLaunchKernelGGL((valu_peak_kernel<float>), dim3(blocks), dim3(256), 0, c->st, c->dbg1.as<float>(), iters, d_clk);
      };
      launch();
      HIPCHK(c, hipStreamSynchronize(c->st));
      HIPCHK(c, hipEventRecord(c->ev[0], c->st));
      for (int r = 0; r < 4; ++r) launch();
      HIPCHK(c, hipEventRecord(c->ev[1], c->st));
      HIPCHK(c, hipStreamSynchronize(c->st));
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
      const double tf = 4.0 * blocks * 256.0 * iters * 16 * 2.0 / (ms * 1e-3) / 1e12;
      long long hclk[2] = {0, 0};
      HIPCHK(c, hipMemcpy(hclk, d_clk, sizeof hclk, hipMemcpyDeviceToHost));
      if (tf > best) {
        best = tf;
        best_cyc = (double)hclk[0] / ((double)iters * 16) / wps;  // cycles per wave-instruction per SIMD
        best_ghz = hclk[1] > 0 ? (double)hclk[0] / ((double)hclk[1] * 10.0) : 0.0;
      }
    }
    *tflops = best;
    if (cycles_per_mfma) *cycles_per_mfma = best_cyc;
    if (clock_ghz) *clock_ghz = best_ghz;
    return 0;
  }
  // variants: accumulators per wave x waves per SIMD; the best one is the ceiling
  for (int nacc : {4, 16})
    for (int wps : {1, 2}) {
      const int blocks = prop.multiProcessorCount * wps;
      auto launch = [&]() {
        if (dtype == GPC_F64) {
          if (nacc == 4)
            hipLaunchKernelGGL((mfma_peak_kernel<double, 4>), dim3(blocks), dim3(256), 0, c->st, c->dbg1.as<double>(), iters, d_clk);
          else
            hipLaunchKernelGGL((mfma_peak_kernel<double, 16>), dim3(blocks), dim3(256), 0, c->st, c->dbg1.as<double>(), iters, d_clk);
        } else {
          if (nacc == 4)
            hipLaunchKernelGGL((mfma_peak_kernel<float, 4>), dim3(blocks), dim3(256), 0, c->st, c->dbg1.as<float>(), iters, d_clk);
          else
            hipLaunchKernelGGL((mfma_peak_kernel<float, 16>), dim3(blocks), dim3(256), 0, c->st, c->dbg1.as<float>(), iters, d_clk);
        }
      };
      launch();
      HIPCHK(c, hipStreamSynchronize(c->st));
      HIPCHK(c, hipEventRecord(c->ev[0], c->st));
      const int reps = 4;
      for (int r = 0; r < reps; ++r) launch();
      HIPCHK(c, hipEventRecord(c->ev[1], c->st));
      HIPCHK(c, hipStreamSynchronize(c->st));
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
      const double flops = (double)reps * blocks * 4.0 * iters * nacc * (2.0 * 16 * 16 * 4);
      const double tf = flops / (ms * 1e-3) / 1e12;
      long long hclk[2] = {0, 0};
      HIPCHK(c, hipMemcpy(hclk, d_clk, sizeof hclk, hipMemcpyDeviceToHost));
      if (tf > best) {
        best = tf;
        best_ghz = hclk[1] > 0 ? (double)hclk[0] / ((double)hclk[1] * 10.0) : 0.0;
        // issue interval of one SIMD, from the aggregate rate and the measured shader clock
        best_cyc = best_ghz * 1e9 * (4.0 * prop.multiProcessorCount) * (2.0 * 16 * 16 * 4) / (tf * 1e12);
      }
    }
  *tflops = best;
  if (cycles_per_mfma) *cycles_per_mfma = best_cyc;
  if (clock_ghz) *clock_ghz = best_ghz;
  return 0;
}

// ------------------------------- test hooks ------------------------------------------

int gpc_debug_gemm(gpc_ctx* c, int dtype, int M, int N, int K, int a_kmajor, int b_kmajor, double alpha,
                   int beta, int klo, int khi, int lower_only, const double* A, const double* B, double* C) {
  if (!c) return -2;
  if (M % TILE || N % TILE || K % TILE || M <= 0 || N <= 0 || K <= 0) FAIL(c, "gpc_debug_gemm: sizes must be multiples of 128");
  if ((lower_only & 1) && M != N) FAIL(c, "gpc_debug_gemm: lower_only needs M == N");
#ifndef GPC_EXPERIMENTS
  if (lower_only & 0x400) FAIL(c, "gpc_debug_gemm: the rectangular tile exists in the experiments build only");
#endif
  HIPCHK(c, hipSetDevice(c->device));
  return dtype == GPC_F64
             ? debug_gemm_impl<double>(c, M, N, K, a_kmajor, b_kmajor, alpha, beta, klo, khi, lower_only, A, B, C)
             : debug_gemm_impl<float>(c, M, N, K, a_kmajor, b_kmajor, alpha, beta, klo, khi, lower_only, A, B, C);
}

int gpc_debug_leaf(gpc_ctx* c, int dtype, const double* A, double* L, double* W, double* logdet, int* info) {
  return gpc_debug_factor(c, dtype, TILE, A, L, W, nullptr, logdet, info);
}

int gpc_debug_factor(gpc_ctx* c, int dtype, int n, const double* A, double* L, double* W, double* Ainv,
                     double* logdet, int* info) {
  if (!c) return -2;
  if (!A || n <= 0) FAIL(c, "gpc_debug_factor: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  return dtype == GPC_F64 ? debug_factor_impl<double>(c, n, A, L, W, Ainv, logdet, info)
                          : debug_factor_impl<float>(c, n, A, L, W, Ainv, logdet, info);
}

// Debug: a 64-bit hash (sum of the bit patterns, wrapping) of every 128 x 128 tile of one workspace matrix of the LAST call
// (which: 0 = A, 1 = W, 2 = T; `sample` = position in the last chunk) -- to find the tile where two schedules differ.
int gpc_debug_workspace_hash(gpc_ctx* c, int dtype, int which, int sample, unsigned long long* out) {
  if (!c || !out || which < 0 || which > 2) return -2;
  HIPCHK(c, hipSetDevice(c->device));
  const int npad = c->npad, nt = npad / TILE;
  DevBuf& m = which == 0 ? c->mA : (which == 1 ? c->mW : c->mT);
  const size_t esz = dtype == GPC_F64 ? 8 : 4;
  if (m.bytes < (size_t)(sample + 1) * npad * npad * esz) FAIL(c, "gpc_debug_workspace_hash: no such sample in the workspace");
  HIPCHK(c, c->dbg3.ensure((size_t)nt * nt * 8));
  if (dtype == GPC_F64)
    hipLaunchKernelGGL((tile_hash_kernel<double>), dim3(nt, nt), dim3(256), 0, c->st, m.as<double>() + (size_t)sample * npad * npad, npad,
                       c->dbg3.as<unsigned long long>());
  else
    hipLaunchKernelGGL((tile_hash_kernel<float>), dim3(nt, nt), dim3(256), 0, c->st, m.as<float>() + (size_t)sample * npad * npad, npad,
                       c->dbg3.as<unsigned long long>());
  HIPCHK(c, hipMemcpyAsync(out, c->dbg3.p, (size_t)nt * nt * 8, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, hipStreamSynchronize(c->st));
  return 0;
}

#ifdef GPC_EXPERIMENTS
// Host-only (no device call): the tile-task graph dag.h derives for a factorization of an npad x npad matrix.
int gpc_debug_dag(int npad, int plan, int nll_blk, int small_tiles, int* counts, int* tasks_out, double* alpha_out,
                  int* succ_out, int cap_tasks, int cap_edges) {
  if (npad <= 0 || npad % TILE || !counts) return -2;
  PlanRecorder rec;
  Factor<double> F;
  F.rec = &rec;
  F.st = nullptr;
  F.batch = 1;
  F.npad = npad;
  F.A = static_cast<double*>(const_cast<void*>(dag_fake_base(0)));
  F.W = static_cast<double*>(const_cast<void*>(dag_fake_base(1)));
  F.Tm = static_cast<double*>(const_cast<void*>(dag_fake_base(2)));
  F.sA = F.sW = F.sT = (long long)npad * npad;
  F.dual_launch = false;
  if (plan == 0) {  // NLL only
    if (nll_blk > 0) {
      F.nll_block = nll_blk;
      F.potrf_nll(0, npad);
    } else {
      F.potrf_inv(0, npad, false, false);
    }
  } else {
    F.potrf_inv(0, npad, true, false);
    if (plan == 1) F.lauum(F.Tm, F.sT);
  }
  const void* bases[3] = {dag_fake_base(0), dag_fake_base(1), dag_fake_base(2)};
  DagPlan P;
  if (!build_dag(rec, bases, npad, sizeof(double), small_tiles, P, true)) return -1;
  counts[0] = P.ntasks;
  counts[1] = (int)P.nedges;
  counts[2] = (int)P.launches.size();
  counts[3] = P.nleaf;
  if (!tasks_out) return 0;
  if (cap_tasks < P.ntasks || cap_edges < (int)P.nedges) return -3;
  for (int t = 0; t < P.ntasks; ++t) {
    const DagTask& d = P.tasks[t];
    const DagPlan::Info& in = P.info[t];
    int* o = tasks_out + (size_t)t * 24;
    o[0] = d.kind == DAG_KIND_LEAF ? 1 : 0;
    o[1] = in.bt;
    o[2] = (d.kind >> 2) & 1;
    o[3] = (d.kind >> 1) & 1;
    o[4] = in.beta;
    const DagPlan::Region* rg[3] = {&in.c, &in.a, &in.b};
    for (int q = 0; q < 3; ++q) {
      o[5 + 5 * q] = rg[q]->buf;
      o[6 + 5 * q] = rg[q]->r0;
      o[7 + 5 * q] = rg[q]->r1;
      o[8 + 5 * q] = rg[q]->c0;
      o[9 + 5 * q] = rg[q]->c1;
    }
    o[20] = d.npred;
    o[21] = d.succ_begin;
    o[22] = d.succ_count;
    o[23] = d.ring;
    alpha_out[t] = in.alpha;
  }
  memcpy(succ_out, P.succ.data(), P.succ.size() * sizeof(int));
  return 0;
}

#endif  // GPC_EXPERIMENTS

}  // extern "C"

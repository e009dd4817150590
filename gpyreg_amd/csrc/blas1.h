// blas1.h -- bandwidth-bound helpers around the factorization (all batched on blockIdx.y):
// triangular matrix-vector products with W = L^-1 (they replace the reference's
// solve_triangular calls at gaussian_process.py:2455-2465), dot products, column sums
// for predict (gaussian_process.py:1747-1764) and small utility copies.
#pragma once
#include "common.h"

namespace gpc {

// z[b][i] = sum_{k<=i} W[b][i][k] * r[b][k].   One wave per row, 4 rows per block.
// The diagonal tile of W is zero above the diagonal, so the row is read up to the end
// of its 128-wide diagonal tile without masking.   grid = (npad/4, batch)
// With row0 > 0 the product is restricted to the diagonal block starting at row0
// (rows row0 + blockIdx.x*4 + wave, columns >= row0): the blocked forward solve.
template <typename T>
__global__ __launch_bounds__(256) void trmv_kernel(const T* __restrict__ W_all, long long sW, int ldw,
                                                   const double* __restrict__ r_all, int npad,
                                                   double* __restrict__ z_all, int row0) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6);
  const T* Wr = W_all + (size_t)b * sW + (size_t)i * ldw;
  const double* r = r_all + (size_t)b * npad;
  const int kend = ((i >> 7) + 1) << 7;
  double s = 0.0;
  for (int k = row0 + lane * VEC; k < kend; k += 64 * VEC) {
    const vec_t wv = *reinterpret_cast<const vec_t*>(Wr + k);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += (double)wv[e] * r[k + e];
  }
  s = wave_sum(s);
  if (lane == 0) z_all[(size_t)b * npad + i] = s;
}

// The same product for the launch that runs UNDER the W^T W GEMM (gpcore.hip: solves beside lauum), where one wave of
// at most 32 VGPRs per SIMD is all that fits beside the GEMM's two resident blocks and every load sees the latency
// of a busy memory system: one wave takes FOUR rows at once, so four loads of W are in flight per load of r.  Per
// row the terms are added in trmv_kernel's order: same bits.   grid = (npad/16, batch), 256 threads
template <typename T>
__global__ __launch_bounds__(256) void trmv_low_kernel(const T* __restrict__ W_all, long long sW, int ldw,
                                                       const double* __restrict__ r_all, int npad,
                                                       double* __restrict__ z_all) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  // everything but the lane's column offset is wave-uniform and lives in scalar registers (the budget is 32 VGPRs):
  // four 16-byte loads in flight (16), the r values (4), four accumulators (8), the offset
  // Beside the GEMM the wave's few fp64 FMAs compete with MFMAs for the DP pipe, and the arbiter serves the older
  // (GEMM) waves first: without a raised priority a row took ~10 us per 128 columns.
  __builtin_amdgcn_s_setprio(3);
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i0 = blockIdx.x * 16 + wv * 4;  // four consecutive rows: one 128-tile, one k-range
  const T* W0 = W_all + (size_t)b * sW + (size_t)i0 * ldw;
  const __amdgpu_buffer_rsrc_t q0 = make_rsrc(W0), q1 = make_rsrc(W0 + ldw), q2 = make_rsrc(W0 + 2 * (size_t)ldw),
                               q3 = make_rsrc(W0 + 3 * (size_t)ldw), qr = make_rsrc(r_all + (size_t)b * npad);
  const int kend = ((i0 >> 7) + 1) << 7;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  static_assert(sizeof(vec_t) == 16, "16-byte loads");
  // one induction variable: the lane's byte offset into r (8 bytes per column); W's is the same (fp64) or half of it
#pragma unroll 1
  for (unsigned orr = lane * VEC * 8u; orr < (unsigned)kend * 8u; orr += 64 * VEC * 8u) {
    const unsigned ow = sizeof(T) == 8 ? orr : orr >> 1;
    const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(q0, ow, 0, 0);
    const u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(q1, ow, 0, 0);
    const u32x4 a2 = __builtin_amdgcn_raw_buffer_load_b128(q2, ow, 0, 0);
    const u32x4 a3 = __builtin_amdgcn_raw_buffer_load_b128(q3, ow, 0, 0);
    const vec_t w0 = *reinterpret_cast<const vec_t*>(&a0), w1 = *reinterpret_cast<const vec_t*>(&a1),
                w2 = *reinterpret_cast<const vec_t*>(&a2), w3 = *reinterpret_cast<const vec_t*>(&a3);
#pragma unroll
    for (int e = 0; e < VEC; e += 2) {  // r: 16 bytes = two doubles per load
      const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(qr, orr + 8u * e, 0, 0);
      const double r0 = reinterpret_cast<const double*>(&rr)[0], r1 = reinterpret_cast<const double*>(&rr)[1];
      s0 += (double)w0[e] * r0;
      s1 += (double)w1[e] * r0;
      s2 += (double)w2[e] * r0;
      s3 += (double)w3[e] * r0;
      s0 += (double)w0[e + 1] * r1;
      s1 += (double)w1[e + 1] * r1;
      s2 += (double)w2[e + 1] * r1;
      s3 += (double)w3[e + 1] * r1;
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  s3 = wave_sum(s3);
  // (the lane index is recomputed here so that it does not occupy a register through the loop)
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
    double* z = z_all + (size_t)b * npad + i0;
    z[0] = s0;
    z[1] = s1;
    z[2] = s2;
    z[3] = s3;
  }
}

// r[b][row0 + i] -= sum_{k < ncols} A[b][row0 + i][col0 + k] * z[b][col0 + k]
// (the off-diagonal update of the blocked forward solve).  grid = (nrows/4, batch)
template <typename T>
__global__ __launch_bounds__(256) void gemv_sub_kernel(const T* __restrict__ A_all, long long sA, int lda,
                                                       const double* __restrict__ z_all, double* __restrict__ r_all,
                                                       int npad, int row0, int col0, int ncols) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6);
  const T* Ar = A_all + (size_t)b * sA + (size_t)i * lda + col0;
  const double* z = z_all + (size_t)b * npad + col0;
  double s = 0.0;
  for (int k = lane * VEC; k < ncols; k += 64 * VEC) {
    const vec_t av = *reinterpret_cast<const vec_t*>(Ar + k);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += (double)av[e] * z[k + e];
  }
  s = wave_sum(s);
  if (lane == 0) r_all[(size_t)b * npad + i] -= s;
}

// W^T z in two deterministic passes.  Pass 1: part[b][ch][k] = sum over the rows of chunk ch
// (TRC rows, only rows >= the diagonal tile of column k) of W[b][i][k] * z[b][i];
// grid = (npad/64, npad/TRC, batch).  Pass 2: out[b][k] = scale[b] * sum_ch part (fixed order).
constexpr int TRC = 128;  // = TILE, so every padded size is a whole number of chunks
// Each lane owns VEC adjacent columns (one 16-byte load per row), each wave every fourth row of the chunk,
// eight loads in flight per lane.   grid = (npad / (64 * VEC), npad / TRC, batch)
// quad != nullptr: block (0, 0, b) also leaves z[b] . z[b] there -- the arithmetic of dot_kernel, without its launch.
// UNR = loads in flight per lane: 8 when the kernel has the chip to itself; 3 (trmv_t_part_low_kernel, <= 32 VGPRs)
// for the launch that runs UNDER the W^T W GEMM (gpcore.hip: solves beside lauum), whose two resident blocks per CU
// leave 32 VGPRs per lane: the rows are added in the same order for every UNR, so the result does not depend on it.
template <typename T, int UNR>
__device__ __forceinline__ void trmv_t_part_body(const T* __restrict__ W_all, long long sW, int ldw,
                                                 const double* __restrict__ z_all, int npad,
                                                 double* __restrict__ part_all, double* __restrict__ quad,
                                                 double (*red)[64 * MM<T>::VEC]) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  const int b = blockIdx.z, lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (quad && blockIdx.x == 0 && blockIdx.y == 0) {
    const double* zz = z_all + (size_t)b * npad;
    double q = 0.0;
    for (int i = threadIdx.x; i < npad; i += 256) q += zz[i] * zz[i];
    q = block_sum_256(q, &red[0][0]);
    if (threadIdx.x == 0) quad[b] = q;
    __syncthreads();
  }
  const int k_raw = (blockIdx.x * 64 + lane) * VEC;
  const int k = min(k_raw, npad - VEC);  // the last block may hang over the edge (npad is a multiple of 128 only)
  const int nch = npad / TRC;
  const int r0 = blockIdx.y * TRC, r1 = r0 + TRC;
  const int istart = max(r0, (int)((blockIdx.x * 64 * VEC) & ~127));
  const T* Wb = W_all + (size_t)b * sW;
  const double* z = z_all + (size_t)b * npad;
  // a block spans 64 * VEC columns: with VEC = 4 (fp32) that is two column tiles, and the chunk that holds the
  // diagonal tile of the first lies ABOVE the diagonal tile of the second (never-written storage): masked per lane
  const bool lane_ok = k_raw < npad && r0 >= (k & ~127);
  double s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.0;
  int i = istart + w;
  for (; i + 4 * (UNR - 1) < r1; i += 4 * UNR) {
    vec_t v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = *reinterpret_cast<const vec_t*>(Wb + (size_t)(i + 4 * u) * ldw + k);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const double zi = z[i + 4 * u];
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += lane_ok ? (double)v[u][e] * zi : 0.0;
    }
  }
  for (; i < r1; i += 4) {
    const vec_t v = *reinterpret_cast<const vec_t*>(Wb + (size_t)i * ldw + k);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] += lane_ok ? (double)v[e] * z[i] : 0.0;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[w][lane * VEC + e] = s[e];
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * VEC; q += 256)
    if (blockIdx.x * 64 * VEC + q < npad)
      part_all[((size_t)b * nch + blockIdx.y) * npad + blockIdx.x * 64 * VEC + q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
}
template <typename T, int UNR = 8>
__global__ __launch_bounds__(256) void trmv_t_part_kernel(const T* __restrict__ W_all, long long sW, int ldw,
                                                          const double* __restrict__ z_all, int npad,
                                                          double* __restrict__ part_all, double* __restrict__ quad) {
  __shared__ double red[4][64 * MM<T>::VEC];
  trmv_t_part_body<T, UNR>(W_all, sW, ldw, z_all, npad, part_all, quad, red);
}
// the form that fits beside two resident 128-tile GEMM blocks per CU: at most 32 VGPRs, three loads in flight per lane
template <typename T>
__global__ __launch_bounds__(256) void trmv_t_part_low_kernel(
    const T* __restrict__ W_all, long long sW, int ldw, const double* __restrict__ z_all, int npad,
    double* __restrict__ part_all, double* __restrict__ quad) {
  __shared__ double red[4][64 * MM<T>::VEC];
  __builtin_amdgcn_s_setprio(3);  // see trmv_low_kernel
  trmv_t_part_body<T, 3>(W_all, sW, ldw, z_all, npad, part_all, quad, red);
}

// grid = (npad/128, batch), 128 threads; scale[b] = 1/sp[b][sp_off] when sp != nullptr
__global__ __launch_bounds__(128) void trmv_t_sum_kernel(const double* __restrict__ part_all, int npad,
                                                         const double* __restrict__ sp_all, int sp_stride,
                                                         int sp_off, double* __restrict__ out_all) {
  const int b = blockIdx.y, k = blockIdx.x * 128 + threadIdx.x;
  const int nch = npad / TRC;
  double v = 0.0;
  for (int ch = k / TRC; ch < nch; ++ch) v += part_all[((size_t)b * nch + ch) * npad + k];
  if (sp_all) v /= sp_all[(size_t)b * sp_stride + sp_off];
  out_all[(size_t)b * npad + k] = v;
}

// out[b] = sum_i x[b][i] * y[b][i].   grid = (1, batch)
__global__ __launch_bounds__(256) void dot_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                  int n, int stride, double* __restrict__ out) {
  __shared__ double sh4[4];
  const int b = blockIdx.y;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += x[(size_t)b * stride + i] * y[(size_t)b * stride + i];
  s = block_sum_256(s, sh4);
  if (threadIdx.x == 0) out[b] = s;
}

// out[b][p] = sum_i Mat[b][i][p] * vec[b][i]   (Mat: n x P row-major; vec stride vstride)
// grid = (P, batch)
__global__ __launch_bounds__(256) void mat_t_vec_kernel(const double* __restrict__ Mat, int n, int P,
                                                        const double* __restrict__ vec, int vstride,
                                                        double* __restrict__ out) {
  __shared__ double sh4[4];
  const int p = blockIdx.x, b = blockIdx.y;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256)
    s += Mat[((size_t)b * n + i) * P + p] * vec[(size_t)b * vstride + i];
  s = block_sum_256(s, sh4);
  if (threadIdx.x == 0) out[(size_t)b * P + p] = s;
}

// The three reductions that close a gradient evaluation in ONE launch (each output by the block, in the order, of the
// kernel it replaces: reduce_parts_kernel, mat_t_vec_kernel x 2).  grid = (P + mean_N + noise_N, batch):
//   x <  P                : gout[b][x]  = sum_tile part[b][tile][x]
//   x <  P + mean_N       : mg[b][p]    = sum_i dm[b][i][p] alpha[b][i]
//   else                  : ng[b][p]    = sum_i dsn2[b][i][p] diagq[b][i]
__global__ __launch_bounds__(256) void grad_tail_kernel(const double* __restrict__ part, int ntiles, int P,
                                                        double* __restrict__ gout, const double* __restrict__ dm,
                                                        int n, int mean_N, const double* __restrict__ alpha,
                                                        double* __restrict__ mg, const double* __restrict__ dsn2,
                                                        int noise_N, const double* __restrict__ diagq,
                                                        double* __restrict__ ng, int vstride) {
  __shared__ double sh4[4];
  const int x = blockIdx.x, b = blockIdx.y;
  double s = 0.0;
  double* out;
  if (x < P) {
    for (int i = threadIdx.x; i < ntiles; i += 256) s += part[((size_t)b * ntiles + i) * P + x];
    out = gout + (size_t)b * P + x;
  } else if (x < P + mean_N) {
    const int p = x - P;
    // (dm == nullptr: the derivative of a constant mean, all ones)
    for (int i = threadIdx.x; i < n; i += 256)
      s += (dm ? dm[((size_t)b * n + i) * mean_N + p] : 1.0) * alpha[(size_t)b * vstride + i];
    out = mg + (size_t)b * mean_N + p;
  } else {
    const int p = x - P - mean_N;
    for (int i = threadIdx.x; i < n; i += 256) s += dsn2[((size_t)b * n + i) * noise_N + p] * diagq[(size_t)b * vstride + i];
    out = ng + (size_t)b * noise_N + p;
  }
  s = block_sum_256(s, sh4);
  if (threadIdx.x == 0) *out = s;
}

// out[b][j] = sum_i A[b][i][j] * Bm[b][i][j]   over rows < nrows.  grid = (mpad/64, batch)
template <typename T>
__global__ __launch_bounds__(256) void colsum_prod_kernel(const T* __restrict__ A_all, long long sA,
                                                          const T* __restrict__ B_all, long long sB,
                                                          int ld, int nrows, int mpad,
                                                          double* __restrict__ out_all) {
  __shared__ double red[4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const T* A = A_all + (size_t)b * sA;
  const T* Bm = B_all + (size_t)b * sB;
  double s = 0.0;
  for (int i = w; i < nrows; i += 4) s += (double)A[(size_t)i * ld + j] * (double)Bm[(size_t)i * ld + j];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0) out_all[(size_t)b * mpad + j] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// out[b][j] = sum_i A[b][i][j] * v[b][i]   (A^T v, A: nrows x mpad).  grid = (mpad/64, batch)
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ A_all, long long sA, int ld,
                                                         const double* __restrict__ v_all, int vstride,
                                                         int nrows, int mpad, double* __restrict__ out_all) {
  __shared__ double red[4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const T* A = A_all + (size_t)b * sA;
  const double* v = v_all + (size_t)b * vstride;
  double s = 0.0;
  for (int i = w; i < nrows; i += 4) s += (double)A[(size_t)i * ld + j] * v[i];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0) out_all[(size_t)b * mpad + j] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// dst[i][j] (double, n x n dense) = j <= i ? src[i][j] : 0       (clean lower factor)
// mode 1: dst = -(symmetrised lower src);  mode 2: plain full copy (low-noise Posterior.L)
// grid = (ceil(n/64), ceil(n/4)), block = (64, 4)
template <typename T>
__global__ void extract_kernel(const T* __restrict__ src, int ld, int n, int mode,
                               double* __restrict__ dst) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= n || j >= n) return;
  double v;
  if (mode == 0)
    v = (j <= i) ? (double)src[(size_t)i * ld + j] : 0.0;
  else if (mode == 1)
    v = -(double)((j <= i) ? src[(size_t)i * ld + j] : src[(size_t)j * ld + i]);
  else
    v = (double)src[(size_t)i * ld + j];
  dst[(size_t)i * n + j] = v;
}

// in place on a padded square buffer: make it the full symmetric NEGATED matrix from
// its lower triangle (the low-noise "L = -inv" of gaussian_process.py:2441-2448).
// grid = (npad/64, npad/4, batch), block = (64, 4)
template <typename T>
__global__ void neg_sym_kernel(T* __restrict__ buf_all, long long sB, int ld, int npad) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= npad || j > i) return;
  T* buf = buf_all + (size_t)blockIdx.z * sB;
  const T v = -buf[(size_t)i * ld + j];
  buf[(size_t)i * ld + j] = v;
  if (j < i) buf[(size_t)j * ld + i] = v;
}

// dst[b][i][j] = (T) src[i][j] for i,j < n else identity   (debug/test upload helper)
template <typename T>
__global__ void pad_load_kernel(const double* __restrict__ src, int n, int npad, T* __restrict__ dst) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= npad || j >= npad) return;
  double v = (i < n && j < n) ? src[(size_t)i * n + j] : ((i == j) ? 1.0 : 0.0);
  dst[(size_t)i * npad + j] = (T)v;
}

// dst[b][r][c] = src[b][r][c] for r < rows, c < cols (rows a multiple of 32, cols of 128; 16-byte aligned rows): the
// L21 blocks of a posterior go back into A with it (plan.h step 6).  A thread moves 16-byte vectors, eight rows in
// flight (round 4; one element per thread moved cfg3's 2.1 GB per update at 1.1 TB/s: 1.9 ms of a 14.9 ms update).
// grid = (ceil(cols / (64 VEC)), rows / 32, batch), block = (64, 4)
template <typename T>
__global__ __launch_bounds__(256) void rect_copy_kernel(const T* __restrict__ src, long long sS, int lds_,
                                                        T* __restrict__ dst, long long sD, int ldd, int rows, int cols) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  const int c = (blockIdx.x * 64 + threadIdx.x) * VEC;
  if (c >= cols) return;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
  const T* s = src + (size_t)blockIdx.z * sS + c;
  T* d = dst + (size_t)blockIdx.z * sD + c;
  vec_t v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const vec_t*>(s + (size_t)(r0 + 4 * u) * lds_);
#pragma unroll
  for (int u = 0; u < 8; ++u) *reinterpret_cast<vec_t*>(d + (size_t)(r0 + 4 * u) * ldd) = v[u];
}

// register-resident MFMA issue loop: the measured ceiling the roofline is quoted against.
// The 4 x 4 accumulator block of the GEMM's k-step, written as one asm block so that the
// compiler cannot put register copies between the MFMAs (a plain C++ loop over an accumulator
// array measured 105 instead of 64 cycles per v_mfma_f64_16x16x4_f64 for that reason;
// a round-1 probe, since removed; profiles/r01g_mfma_cadence.txt).  NACC is kept as a template parameter for the C ABI's variants:
// every variant issues the same 16 independent MFMAs per iteration.
#define GPC_PEAK_M(op, c, a, b) op " %" #c ", %" #a ", %" #b ", %" #c "\n"
#define GPC_PEAK_BLOCK(op)                                                                                        \
  asm volatile(GPC_PEAK_M(op, 0, 16, 20) GPC_PEAK_M(op, 1, 16, 21) GPC_PEAK_M(op, 2, 16, 22) GPC_PEAK_M(op, 3, 16, 23)   \
               GPC_PEAK_M(op, 4, 17, 20) GPC_PEAK_M(op, 5, 17, 21) GPC_PEAK_M(op, 6, 17, 22) GPC_PEAK_M(op, 7, 17, 23)   \
               GPC_PEAK_M(op, 8, 18, 20) GPC_PEAK_M(op, 9, 18, 21) GPC_PEAK_M(op, 10, 18, 22) GPC_PEAK_M(op, 11, 18, 23) \
               GPC_PEAK_M(op, 12, 19, 20) GPC_PEAK_M(op, 13, 19, 21) GPC_PEAK_M(op, 14, 19, 22) GPC_PEAK_M(op, 15, 19, 23) \
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),        \
                 "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])  \
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3))
template <typename T, int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(T* out, int iters, long long* clk) {
  using acc_t = typename MM<T>::acc_t;
  acc_t c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = acc_t{0, 0, 0, 0};
  const T a0 = (T)(threadIdx.x * 1e-3), a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3;
  const T b0 = (T)(1.0 + threadIdx.x * 1e-4), b1 = b0 + (T)1, b2 = b0 + (T)2, b3 = b0 + (T)3;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters * NACC / 16; ++it) {
    if constexpr (sizeof(T) == 8)
      GPC_PEAK_BLOCK("v_mfma_f64_16x16x4_f64");
    else
      GPC_PEAK_BLOCK("v_mfma_f32_16x16x4_f32");
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;  // shader-clock ticks for iters*NACC MFMAs of this wave
    clk[1] = w1 - w0;  // 100 MHz reference ticks
  }
  T s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
#undef GPC_PEAK_BLOCK
#undef GPC_PEAK_M

// ---- rank-one append of a training point to resident posteriors (gaussian_process.py:750-844) ----
// dst[b] ((npn x npn), identity-padded) <- src[b] (np x np top-left block).  grid = (npn/64, npn/4, batch)
template <typename T>
__global__ void grow_copy_kernel(const T* __restrict__ src, int np, T* __restrict__ dst, int npn) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= npn || j >= npn) return;
  const size_t b = blockIdx.z;
  T v = (i == j) ? (T)1 : (T)0;
  if (i < np && j < np) v = src[b * (size_t)np * np + (size_t)i * np + j];
  dst[b * (size_t)npn * npn + (size_t)i * npn + j] = v;
}

// Row n of the lower factor and of its inverse, and the updated alpha, for every sample b:
//   Lo[n][k] = l[k] * cl[b] (k < n),  Lo[n][n] = dl[b]
//   W [n][k] = au[k] * cw[b] (k < n), W [n][n] = 1 / dl[b]
//   alpha[k] += ca[b] * au[k] * cau[b] (k < n),  alpha[n] = -ca[b]
// l = W Ks, au = W^T l (un-normalised); the host derived the per-sample coefficients.
// coef[b] = {cl, dl, cw, ca, cau}.   grid = (npad/256, batch)
template <typename T>
__global__ __launch_bounds__(256) void append_row_kernel(T* __restrict__ A_all, T* __restrict__ W_all, long long sM,
                                                         int ld, int n, const double* __restrict__ l_all,
                                                         const double* __restrict__ au_all, int npad,
                                                         const double* __restrict__ coef,
                                                         double* __restrict__ alpha_all) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k > n) return;
  const double* cf = coef + (size_t)b * 5;
  T* Arow = A_all + (size_t)b * sM + (size_t)n * ld;
  T* Wrow = W_all + (size_t)b * sM + (size_t)n * ld;
  double* alpha = alpha_all + (size_t)b * npad;
  if (k < n) {
    const double lv = l_all[(size_t)b * npad + k], av = au_all[(size_t)b * npad + k];
    Arow[k] = (T)(lv * cf[0]);
    Wrow[k] = (T)(av * cf[2]);
    alpha[k] += cf[3] * av * cf[4];
  } else {
    Arow[n] = (T)cf[1];
    Wrow[n] = (T)(1.0 / cf[1]);
    alpha[n] = -cf[3];
  }
}

// Low-noise parametrisation (Posterior.L = -(K + Sigma)^-1, full symmetric; gaussian_process.py:819-827):
// with v = -au * cv (cv = 1 / v_star),
//   L[i][j] += v[i] * au[j] (i, j < n);  L[i][n] = L[n][i] = -v[i];  L[n][n] = -cv;
//   alpha[i] += ca * au[i],  alpha[n] = -ca.        coef = {cv, -, -, ca, -}
// grid = ((n + 64) / 64, (n + 4) / 4), block = (64, 4): one thread per entry of the (n+1) x (n+1) block.
template <typename T>
__global__ void append_low_kernel(T* __restrict__ A, int ld, int n, const double* __restrict__ au,
                                  const double* __restrict__ coef, double* __restrict__ alpha) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i > n || j > n) return;
  const double cv = coef[0], ca = coef[3];
  T* p = A + (size_t)i * ld + j;
  if (i < n && j < n) {
    *p = (T)((double)*p - au[i] * cv * au[j]);
  } else if (i == n && j == n) {
    *p = (T)(-cv);
    alpha[n] = -ca;
  } else {
    const int k = i < n ? i : j;
    *p = (T)(au[k] * cv);  // -v[k]
  }
  if (j == 0 && i < n) alpha[i] += ca * au[i];
}

// ---- caller-provided covariance (gpc_nll_batch_K / gpc_posterior_batch_K / gpc_predict_K) ----
// A (npad x npad, identity padding) = K / sp[2] + diag(dvec)   from a dense n x n double matrix
// (sp[2] = SP_KSCALE: gaussian_process.py:2416 / :2432).   grid = (npad/64, npad/4), block = (64, 4)
template <typename T>
__global__ void load_K_kernel(const double* __restrict__ K, int n, int npad, const double* __restrict__ sp,
                              const double* __restrict__ dvec, T* __restrict__ A) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= npad || j >= npad) return;
  double v = (i == j) ? 1.0 : 0.0;
  if (i < n && j < n) {
    v = K[(size_t)i * n + j] / sp[2];
    if (i == j) v += dvec[i];
  }
  A[(size_t)i * npad + j] = (T)v;
}

// dst (rpad x cpad, zero padding) = (T) src (r x c doubles).   grid = (cpad/64, rpad/4), block = (64, 4)
template <typename T>
__global__ void pad_rect_kernel(const double* __restrict__ src, int r, int c, int rpad, int cpad,
                                T* __restrict__ dst) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= rpad || j >= cpad) return;
  dst[(size_t)i * cpad + j] = (T)((i < r && j < c) ? src[(size_t)i * c + j] : 0.0);
}

// diagQ[b][i] = Ainv[b][i][i] / sl - alpha_i^2 and out[b * P + P - 1] = trace(Q)  (sp[3] = SP_SL).
// grid = (1, batch), 256 threads
template <typename T>
__global__ __launch_bounds__(256) void diagq_kernel(const T* __restrict__ Ainv_all, long long sA, int ld,
                                                    const double* __restrict__ alpha_all,
                                                    const double* __restrict__ sp_all, int n,
                                                    double* __restrict__ diagq_all, double* __restrict__ out, int P) {
  __shared__ double sh4[4];
  const int b = blockIdx.y;
  const T* Ainv = Ainv_all + (size_t)b * sA;
  const double* alpha = alpha_all + (size_t)b * ld;
  const double invsl = 1.0 / sp_all[(size_t)b * 4 + 3];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double q = (double)Ainv[(size_t)i * ld + i] * invsl - alpha[i] * alpha[i];
    diagq_all[(size_t)b * ld + i] = q;
    s += q;
  }
  s = block_sum_256(s, sh4);
  if (threadIdx.x == 0) out[(size_t)b * P + P - 1] = s;
}

// part[block] = sum over this block's entries of Q_ij * dK_ij, Q = Ainv / sl - alpha alpha^T taken from the
// LOWER triangle of Ainv (Q is symmetric), dK a dense n x n plane (gaussian_process.py:2487-2488: the
// reference sums the full matrices).  Fixed grid-stride assignment: deterministic.  grid = (nblk), 256 threads
template <typename T>
__global__ __launch_bounds__(256) void trace_plane_kernel(const T* __restrict__ Ainv, int ld,
                                                          const double* __restrict__ alpha,
                                                          const double* __restrict__ sp,
                                                          const double* __restrict__ dK, int n,
                                                          double* __restrict__ part) {
  __shared__ double sh4[4];
  const double invsl = 1.0 / sp[3];
  const long long tot = (long long)n * n;
  double s = 0.0;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / n), j = (int)(idx - (long long)i * n);
    const double a = (double)(j <= i ? Ainv[(size_t)i * ld + j] : Ainv[(size_t)j * ld + i]);
    s += (a * invsl - alpha[i] * alpha[j]) * dK[idx];
  }
  s = block_sum_256(s, sh4);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// independent-chain VALU FMA loop (what the non-MFMA kernels are bounded by)
template <typename T>
__global__ __launch_bounds__(256) void valu_peak_kernel(T* out, int iters, long long* clk) {
  T acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (T)(threadIdx.x + i);
  const T a = (T)(1.0 + threadIdx.x * 1e-9), bq = (T)(threadIdx.x * 1e-7);
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc[i] * a + bq;
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
  T s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace gpc

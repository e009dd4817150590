// covfun.h -- covariance functions on the device: kernel-matrix build, the fused
// gradient contraction  sum_ij Q_ij dK_ij/dtheta  (dK is never materialised), cross
// covariances for predict, and the standalone compute() used by the plugin API.
//
// Reference formulas (cited lines are in the reference checkout):
//   SE      covariance_functions.py:158-183   K = sf2 exp(-r2/2);  dK_l = K d_l;  dK_sf = 2K
//   Matern  covariance_functions.py:210-218,244-282  t = sqrt(r2) of sqrt(nu)/ell-scaled X
//           K = sf2 f(t) e^-t;  dK_l = sf2 (df(t) e^-t) d_l   (nu=1: df = 1/t -> NaN on diag)
//   RQ      covariance_functions.py:324-363   M = 1 + r2/(2a); K = sf2 M^-a;
//           dK_l = sf2 M^(-a-1) d_l;  dK_a = K (r2/(2M) - a log M)
//   iso     isotropic_covariance_functions.py:127-158, :196-218  (single ell; dK_l = F * r2)
// with d_l = (xs_i[l] - xs_j[l])^2 on the SCALED inputs xs and r2 = sum_l d_l summed in
// ascending l (the order scipy's pdist uses, which keeps results within rounding of it).
#pragma once
#include "common.h"

namespace gpc {

enum { K_SE = 0, K_MATERN = 1, K_RQ = 2, K_SE_ISO = 3, K_MATERN_ISO = 4 };

// per-sample scalars, double, SP_STRIDE apart
enum { SP_SF2 = 0, SP_RQA = 1, SP_KSCALE = 2, SP_SL = 3, SP_STRIDE = 4 };

struct CovDesc {
  int kind;    // K_*
  int degree;  // Matern: 1/3/5
  int D;
  int cov_N;
};

inline __host__ __device__ bool cov_is_iso(int kind) { return kind == K_SE_ISO || kind == K_MATERN_ISO; }

struct PairVal {
  double K;   // covariance value
  double F;   // dK/dlog(ell_l) = F * d_l   (iso: F * r2)
  double Ka;  // RQ only: dK/dlog(alpha)
};

__device__ __forceinline__ PairVal pair_eval(int kind, int degree, double r2, double sf2, double rqa) {
  PairVal o;
  o.Ka = 0.0;
  if (kind == K_SE || kind == K_SE_ISO) {
    o.K = sf2 * exp(-r2 / 2);
    o.F = o.K;
  } else if (kind == K_MATERN || kind == K_MATERN_ISO) {
    const double t = sqrt(r2);
    const double e = exp(-t);
    double f, df;
    if (degree == 1) {
      f = 1.0;
      df = 1.0 / t;
    } else if (degree == 3) {
      f = 1.0 + t;
      df = 1.0;
    } else {
      // multiplications by 1/3 (<= 1 ulp from the reference's divisions): an fp64 division is a
      // ~15-instruction sequence and these kernels are VALU bound, not HBM bound
      constexpr double third = 1.0 / 3.0;
      f = 1.0 + t * (1.0 + t * third);
      df = (1.0 + t) * third;
    }
    o.K = sf2 * f * e;
    o.F = sf2 * (df * e);
  } else {
    const double Mv = 1.0 + r2 * (0.5 / rqa);  // the quotient is loop invariant
    o.K = sf2 * pow(Mv, -rqa);
    o.F = sf2 * pow(Mv, -rqa - 1.0);
    o.Ka = o.K * (0.5 * r2 / Mv - rqa * log(Mv));
  }
  return o;
}

// Xs[b][i][h] = X[i][h] * mul[b][h] / div[b][h]  (rows >= n are zero padding)
__global__ void scale_x_kernel(const double* __restrict__ X, int n, int npad, int D,
                               const double* __restrict__ mul, const double* __restrict__ dv,
                               double* __restrict__ Xs) {
  const int b = blockIdx.y;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)npad * D) return;
  const int i = (int)(idx / D), h = (int)(idx % D);
  double v = 0.0;
  if (i < n) v = X[(size_t)i * D + h] * mul[(size_t)b * D + h] / dv[(size_t)b * D + h];
  Xs[(size_t)b * npad * D + idx] = v;
}

constexpr int CT = 64;   // tile of the elementwise covariance kernels
constexpr int DCH = 32;  // input dimensions staged in LDS per pass

// stage rows [r0, r0+64) of Xs, dims [h0, h0+dc) into sh[64][DCH+1]
__device__ __forceinline__ void stage_x(double (*sh)[DCH + 1], const double* __restrict__ Xs, int D,
                                        int r0, int h0, int dc, int t) {
  const int r = t >> 2;  // 4 threads per row, no integer division by the runtime chunk width
  for (int h = t & 3; h < dc; h += 4) sh[r][h] = Xs[(size_t)(r0 + r) * D + h0 + h];
}

// ---------------------------------------------------------------------------------
// Pair evaluation specialised at compile time on (kernel family, Matern degree): the N^2 passes
// below are fp64-VALU bound, so the family switch is taken once per launch (launch_build /
// launch_trace), the square root is the hardware rsq estimate + one Newton step + one
// correction (<= 1 ulp, 9 instructions instead of the ~20 of a correctly rounded division-based
// sqrt) and the rational quadratic needs one log, one exp and one reciprocal instead of two pows
// and a log:  M^-a = exp(-a log M),  M^(-a-1) = M^-a / M.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double sqrt_fast(double x) {
  double r = __builtin_amdgcn_rsq(x);      // ~5e-8 relative; +inf at 0
  r = r * fma(-0.5 * x, r * r, 1.5);        // one Newton step: ~4e-15
  double t = x * r;
  const double e = fma(-t, t, x);           // residual x - t^2
  t = fma(e, 0.5 * r, t);                   // t + e / (2 sqrt(x)): <= 1 ulp
  return x > 0.0 ? t : 0.0;
}
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// exp(x) for the arguments of the covariance functions (x <= 0 up to rounding; any finite x works,
// overflow is not handled).  Cody-Waite reduction x = n ln2 + r, |r| <= ln2 / 2, Taylor polynomial of
// degree 13 (truncation 4e-18), scaling by ldexp (exact; gradual underflow to 0 for x < -745).
// The library exp costs ~46 instructions per call in these kernels: its 16 unrolled copies re-materialise
// the fp64 polynomial coefficients with v_mov pairs each time (358 v_mov_b32 in the old trace kernel);
// here the coefficients are read once from constant memory into scalar registers and every Horner
// step is one v_fma_f64 with an SGPR-pair operand: 19 instructions, < 1 ulp from the library's result.
__constant__ double GPC_EXPC[14] = {1.0,
                                    1.0,
                                    1.0 / 2,
                                    1.0 / 6,
                                    1.0 / 24,
                                    1.0 / 120,
                                    1.0 / 720,
                                    1.0 / 5040,
                                    1.0 / 40320,
                                    1.0 / 362880,
                                    1.0 / 3628800,
                                    1.0 / 39916800,
                                    1.0 / 479001600,
                                    1.0 / 6227020800.0};
struct ExpC {
  double c[14];
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int i = 0; i < 14; ++i) c[i] = GPC_EXPC[i];
  }
  __device__ __forceinline__ double operator()(double x) const {
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, x);  // ln2 high part (trailing zeros: n * hi exact)
    r = fma(-n, 1.90821492927058770002e-10, r);         // ln2 low part
    double p = c[13];
#pragma unroll
    for (int i = 12; i >= 0; --i) p = fma(p, r, c[i]);
    return __builtin_ldexp(p, (int)n);
  }
};

template <int KIND, int DEG>
__device__ __forceinline__ PairVal pair_eval_t(double r2, double sf2, double rqa, const ExpC& ex) {
  PairVal o;
  o.Ka = 0.0;
  if constexpr (KIND == K_SE || KIND == K_SE_ISO) {
    o.K = sf2 * ex(-0.5 * r2);
    o.F = o.K;
  } else if constexpr (KIND == K_MATERN || KIND == K_MATERN_ISO) {
    const double t = sqrt_fast(r2);
    const double e = sf2 * ex(-t);
    if constexpr (DEG == 1) {
      o.K = e;
      o.F = e / t;  // 1/t: +inf on the diagonal, by the reference's definition (:276-279)
    } else if constexpr (DEG == 3) {
      o.K = fma(t, e, e);
      o.F = e;
    } else {
      constexpr double third = 1.0 / 3.0;
      o.K = e * fma(t, fma(t, third, 1.0), 1.0);
      o.F = e * (fma(t, third, third));
    }
  } else {
    const double Mv = fma(r2, 0.5 / rqa, 1.0);
    const double lM = log(Mv);
    const double rM = rcp_fast(Mv);
    o.K = sf2 * ex(-rqa * lM);
    o.F = o.K * rM;
    o.Ka = o.K * fma(0.5 * r2, rM, -rqa * lM);
  }
  return o;
}

// squared distances of the 4 x 4 pairs of one thread (rows ty + 16a of tile i, rows tx + 16c of tile j),
// dimensions summed in ascending order; leaves the LAST staged chunk of dimensions in xi / xj
__device__ __forceinline__ void tile_r2_ab(double (&r2)[4][4], double (*xi)[DCH + 1], double (*xj)[DCH + 1],
                                           const double* __restrict__ Xa, const double* __restrict__ Xb, int D, int i0,
                                           int j0, int t, int tx, int ty) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) r2[a][c] = 0.0;
  for (int h0 = 0; h0 < D; h0 += DCH) {
    const int dc = min(DCH, D - h0);
    __syncthreads();
    stage_x(xi, Xa, D, i0, h0, dc, t);
    stage_x(xj, Xb, D, j0, h0, dc, t);
    __syncthreads();
    for (int h = 0; h < dc; ++h) {
      double vi[4], vj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) vi[a] = xi[ty + 16 * a][h];
#pragma unroll
      for (int c = 0; c < 4; ++c) vj[c] = xj[tx + 16 * c][h];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double d = vi[a] - vj[c];
          r2[a][c] = fma(d, d, r2[a][c]);
        }
    }
  }
}

__device__ __forceinline__ void tile_r2(double (&r2)[4][4], double (*xi)[DCH + 1], double (*xj)[DCH + 1],
                                        const double* __restrict__ Xs, int D, int i0, int j0, int t, int tx,
                                        int ty) {
  tile_r2_ab(r2, xi, xj, Xs, Xs, D, i0, j0, t, tx, ty);
}

__device__ __forceinline__ void lower_tile(int bx, int& ti, int& tj) {
  int i = (int)((sqrtf(8.f * (float)bx + 1.f) - 1.f) * 0.5f);
  while (i * (i + 1) / 2 > bx) --i;
  while ((i + 1) * (i + 2) / 2 <= bx) ++i;
  ti = i;
  tj = bx - i * (i + 1) / 2;
}

// Sum four per-lane values over the 64 lanes of a wave with 7 exchanges instead of 24: two
// halving steps leave each lane one value (lane bits 5, 4 select which), four more finish it.
// On return lane l holds the wave total of v[(l >> 4) & 3].  Fixed order: deterministic.
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3, int lane) {
  const bool b5 = lane & 32, b4 = lane & 16;
  double k0 = b5 ? v2 : v0, k1 = b5 ? v3 : v1;
  k0 += __shfl_xor(b5 ? v0 : v2, 32, 64);
  k1 += __shfl_xor(b5 ? v1 : v3, 32, 64);
  double u = (b4 ? k1 : k0) + __shfl_xor(b4 ? k0 : k1, 16, 64);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) u += __shfl_xor(u, o, 64);
  return u;
}


// ---------------------------------------------------------------------------------
// fp32 mode (round 4): the same two N^2 passes with a NATIVE fp32 functor.  In fp32 mode the matrix is stored,
// factored and inverted in fp32 anyway (relative error 6e-8 per entry before the factorization amplifies it), so
// evaluating exp / log / pow in fp64 VALU (~19 / ~40 instructions each, 4 cycles per wave and instruction) bought no
// accuracy and made the passes VALU-bound at 6-8 % of HBM peak on cfg4 (N = 16384, D = 20, rational quadratic).
// Here:  * the scaled inputs are staged in LDS as floats and read two DIMENSIONS at a time (ds_read_b64), the
//          differences and squares are packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two dimensions per lane and
//          instruction, the fp32 vector peak), even and odd dimensions summed separately and added at the end --
//          direct differences, never |x|^2 + |y|^2 - 2 x.y: near-duplicate points keep their relative accuracy;
//        * exp, log, reciprocal and square root are the hardware's v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32
//          (~1 ulp, 8 cycles per wave each) instead of fp64 polynomial code;
//        * per-thread partial sums over the thread's 16 pairs are fp32, everything across lanes, waves and tiles
//          (wave_sum4, reduce_parts_kernel) stays fp64 and in a fixed order.
// The reference has no fp32 path; the bar is north_star's 1e-3 on nlZ and gradient.
// ---------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int XLD32 = DCH + 4;  // LDS row stride in floats: 16 rows x 36 floats hit 16 distinct bank quads (ds_read_b64)

struct PairVal32 {
  float K, F, Ka;
};

template <int KIND, int DEG>
__device__ __forceinline__ PairVal32 pair_eval32(float r2, float sf2, float rqa, float half_over_a) {
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  PairVal32 o;
  o.Ka = 0.f;
  if constexpr (KIND == K_SE || KIND == K_SE_ISO) {
    o.K = sf2 * __builtin_amdgcn_exp2f(r2 * (-0.5f * LOG2E));
    o.F = o.K;
  } else if constexpr (KIND == K_MATERN || KIND == K_MATERN_ISO) {
    const float t = __builtin_amdgcn_sqrtf(r2);
    const float e = sf2 * __builtin_amdgcn_exp2f(t * -LOG2E);
    if constexpr (DEG == 1) {
      o.K = e;
      o.F = e * __builtin_amdgcn_rcpf(t);  // +inf on the diagonal, as the reference defines it (:276-279)
    } else if constexpr (DEG == 3) {
      o.K = fmaf(t, e, e);
      o.F = e;
    } else {
      constexpr float third = 1.0f / 3.0f;
      o.K = e * fmaf(t, fmaf(t, third, 1.0f), 1.0f);
      o.F = e * fmaf(t, third, third);
    }
  } else {
    const float Mv = fmaf(r2, half_over_a, 1.0f);
    const float l2 = __builtin_amdgcn_logf(Mv);  // log2
    const float rM = __builtin_amdgcn_rcpf(Mv);
    o.K = sf2 * __builtin_amdgcn_exp2f(-rqa * l2);
    o.F = o.K * rM;
    o.Ka = o.K * fmaf(0.5f * r2, rM, -(rqa * LN2) * l2);
  }
  return o;
}

// rows [r0, r0+64), dims [h0, h0+dc) of Xs (double) -> sh[64][XLD32] floats; dims dc .. dc rounded up to even are 0
__device__ __forceinline__ void stage_x32(float* __restrict__ sh, const double* __restrict__ Xs, int D, int r0, int h0,
                                          int dc, int t) {
  const int r = t >> 2;
  const int dce = (dc + 1) & ~1;
  for (int h = t & 3; h < dce; h += 4) sh[r * XLD32 + h] = h < dc ? (float)Xs[(size_t)(r0 + r) * D + h0 + h] : 0.f;
}

// squared distances of the thread's 4 x 4 pairs, two dimensions per packed instruction; leaves the last chunk staged
__device__ __forceinline__ void tile_r2_32(float (&r2)[4][4], float* __restrict__ xi, float* __restrict__ xj,
                                           const double* __restrict__ Xs, int D, int i0, int j0, int t, int tx, int ty) {
  f32x2 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x2{0.f, 0.f};
  for (int h0 = 0; h0 < D; h0 += DCH) {
    const int dc = min(DCH, D - h0);
    __syncthreads();
    stage_x32(xi, Xs, D, i0, h0, dc, t);
    stage_x32(xj, Xs, D, j0, h0, dc, t);
    __syncthreads();
    for (int h = 0; h < dc; h += 2) {
      f32x2 vi[4], vj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) vi[a] = *reinterpret_cast<const f32x2*>(xi + (ty + 16 * a) * XLD32 + h);
#pragma unroll
      for (int c = 0; c < 4; ++c) vj[c] = *reinterpret_cast<const f32x2*>(xj + (tx + 16 * c) * XLD32 + h);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2 d = vi[a] - vj[c];
          acc[a][c] = __builtin_elementwise_fma(d, d, acc[a][c]);
        }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) r2[a][c] = acc[a][c].x + acc[a][c].y;
}

template <int KIND, int DEG>
__device__ __forceinline__ void build_tile32(const CovDesc& cd, const double* __restrict__ Xs_all,
                                             const double* __restrict__ sp_all, const double* __restrict__ dvec_all,
                                             int n, int npad, float* __restrict__ A_all, long long sA, int lda,
                                             int tile, int b, int bs, float* __restrict__ xi, float* __restrict__ xj) {
  // bs: the sample index into sp_all / dvec_all (= b, except where the caller holds one sample's copies: bs = 0)
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  int ti, tj;
  lower_tile(tile, ti, tj);
  const int i0 = ti * CT, j0 = tj * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* sp = sp_all + (size_t)bs * SP_STRIDE;
  const double* dvec = dvec_all + (size_t)bs * npad;
  float* A = A_all + (size_t)b * sA;
  float r2[4][4];
  tile_r2_32(r2, xi, xj, Xs, cd.D, i0, j0, t, tx, ty);
  const float sfs = (float)(sp[SP_SF2] / sp[SP_KSCALE]);  // K / (sn2_div * sn2_mult), :2416
  const float rqa = (float)sp[SP_RQA], hoa = (float)(0.5 / sp[SP_RQA]);
  // straight-line code for the 16 pairs (r2 is finite everywhere, padding included: selects, no branches); the diagonal
  // -- noise term, identity padding -- exists in diagonal tiles only, a block-uniform branch
  float v[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
      const float k = pair_eval32<KIND, DEG>(r2[a][c], sfs, rqa, hoa).K;
      v[a][c] = (i < n && j < n) ? k : 0.0f;
    }
  if (ti == tj) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
        if (i == j) v[a][c] = i < n ? (float)((double)v[a][c] + dvec[i]) : 1.0f;
      }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) A[(size_t)(i0 + ty + 16 * a) * lda + j0 + tx + 16 * c] = v[a][c];
}

template <int KIND, int DEG>
__device__ __forceinline__ void trace_tile32(const CovDesc& cd, const double* __restrict__ Xs_all,
                                             const double* __restrict__ sp_all, const double* __restrict__ alpha_all,
                                             int n, int npad, const float* __restrict__ Kinv_all, long long sK, int ldk,
                                             double* __restrict__ part_all, int ntiles, double* __restrict__ diagQ_all,
                                             float* __restrict__ xi, float* __restrict__ xj, double* __restrict__ wpart) {
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, b = blockIdx.y;
  const int lane = t & 63, w = t >> 6;
  const int P = cd.cov_N + 1;
  const int P4 = ((P + 3) & ~3) + 4;
  constexpr bool ISO = (KIND == K_SE_ISO || KIND == K_MATERN_ISO);
  int ti, tj;
  lower_tile(blockIdx.x, ti, tj);
  const int i0 = ti * CT, j0 = tj * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* sp = sp_all + (size_t)b * SP_STRIDE;
  const double* alpha = alpha_all + (size_t)b * npad;
  const float* Kinv = Kinv_all + (size_t)b * sK;
  double* part = part_all + ((size_t)b * ntiles + blockIdx.x) * P;

  float kin[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) kin[a][c] = Kinv[(size_t)(i0 + ty + 16 * a) * ldk + j0 + tx + 16 * c];
  float r2[4][4];
  tile_r2_32(r2, xi, xj, Xs, cd.D, i0, j0, t, tx, ty);
  float al_i[4], al_j[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    al_i[a] = (float)alpha[i0 + ty + 16 * a];
    al_j[a] = (float)alpha[j0 + tx + 16 * a];
  }
  const float sf2 = (float)sp[SP_SF2], rqa = (float)sp[SP_RQA], hoa = (float)(0.5 / sp[SP_RQA]);
  const float invsl = (float)(1.0 / sp[SP_SL]);
  f32x2 qF[4][4];  // (qF, qF): both dimensions of a packed step are weighted by the pair's factor
  float g_sf = 0.f, g_a = 0.f, g_iso = 0.f, trq = 0.f;
  // (one branch per pair, as in the fp64 kernel: evaluating the 16 pairs as straight-line code with selects was
  // measured -- the compiler then keeps all 16 evaluations in flight, 172 instead of 110 VGPRs, two waves per SIMD
  // instead of four, 0.61 instead of 0.44 ms on cfg4)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
      const bool valid = (i < n) && (j <= i);
      float qf = 0.f;
      if (valid) {
        const float Q = fmaf(kin[a][c], invsl, -al_i[a] * al_j[c]);
        const float qw = (i == j) ? Q : 2.0f * Q;
        const PairVal32 pv = pair_eval32<KIND, DEG>(r2[a][c], sf2, rqa, hoa);
        g_sf = fmaf(qw, 2.0f * pv.K, g_sf);
        if constexpr (KIND == K_RQ) g_a = fmaf(qw, pv.Ka, g_a);
        qf = qw * pv.F;
        if constexpr (ISO) g_iso = fmaf(qf, r2[a][c], g_iso);
        if (i == j) {
          trq += Q;
          diagQ_all[(size_t)b * npad + i] = (double)Q;
        }
      }
      qF[a][c] = f32x2{qf, qf};  // masked entries exactly 0; a Matern-1 diagonal entry is inf * 0 = NaN, as in the reference
    }

  const int own = lane >> 4;
  if constexpr (ISO) {
    const double u = wave_sum4((double)g_iso, (double)g_sf, (double)trq, 0.0, lane);
    if ((lane & 15) == 0 && own < 3) wpart[w * P4 + own] = u;
  } else {
    // second sweep over the input dimensions, four per reduction: G_h = sum_e qF[e] d_h[e]^2
    for (int h0 = 0; h0 < cd.D; h0 += DCH) {
      const int dc = min(DCH, cd.D - h0);
      if (cd.D > DCH) {
        __syncthreads();
        stage_x32(xi, Xs, cd.D, i0, h0, dc, t);
        stage_x32(xj, Xs, cd.D, j0, h0, dc, t);
        __syncthreads();
      }
      for (int h = 0; h < dc; h += 4) {
        f32x2 s2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int hh = h + 2 * q;
          if (hh >= dc) break;  // block-uniform
          f32x2 vi[4], vj[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) vi[a] = *reinterpret_cast<const f32x2*>(xi + (ty + 16 * a) * XLD32 + hh);
#pragma unroll
          for (int c = 0; c < 4; ++c) vj[c] = *reinterpret_cast<const f32x2*>(xj + (tx + 16 * c) * XLD32 + hh);
          f32x2 s = f32x2{0.f, 0.f};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x2 d = vi[a] - vj[c];
              s = __builtin_elementwise_fma(d * d, qF[a][c], s);
            }
          s2[q] = s;
        }
        const double u = wave_sum4((double)s2[0].x, (double)s2[0].y, (double)s2[1].x, (double)s2[1].y, lane);
        if ((lane & 15) == 0 && h + own < dc) wpart[w * P4 + h0 + h + own] = u;
      }
    }
    const double u = wave_sum4((double)g_sf, (double)g_a, (double)trq, 0.0, lane);
    if ((lane & 15) == 0) {
      if (own == 0) wpart[w * P4 + cd.D] = u;
      if (own == 1 && KIND == K_RQ) wpart[w * P4 + cd.D + 1] = u;
      if (own == 2) wpart[w * P4 + P - 1] = u;
    }
  }
  __syncthreads();
  for (int p = t; p < P; p += 256)
    part[p] = wpart[p] + wpart[P4 + p] + wpart[2 * P4 + p] + wpart[3 * P4 + p];
}

// ---------------------------------------------------------------------------------
// A[b] (lower 64x64 tiles) = K(Xs, Xs) / sp[SP_KSCALE] + diag(dvec), identity in the padding.
// grid = (lower tiles of npad/64, batch)
// ---------------------------------------------------------------------------------
template <typename T, int KIND, int DEG>
__device__ __forceinline__ void build_tile_at(const CovDesc& cd, const double* __restrict__ Xs_all,
                                              const double* __restrict__ sp_all, const double* __restrict__ dvec_all,
                                              int n, int npad, T* __restrict__ A_all, long long sA, int lda, int tile,
                                              int b, int bs, double (*xi)[DCH + 1], double (*xj)[DCH + 1]) {
  if constexpr (sizeof(T) == 4) {  // fp32 mode: the native fp32 functor (the LDS arrays are reused as float images)
    build_tile32<KIND, DEG>(cd, Xs_all, sp_all, dvec_all, n, npad, A_all, sA, lda, tile, b, bs,
                            reinterpret_cast<float*>(&xi[0][0]), reinterpret_cast<float*>(&xj[0][0]));
    return;
  }
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  int ti, tj;
  lower_tile(tile, ti, tj);
  const int i0 = ti * CT, j0 = tj * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* sp = sp_all + (size_t)bs * SP_STRIDE;
  const double* dvec = dvec_all + (size_t)bs * npad;
  T* A = A_all + (size_t)b * sA;

  double r2[4][4];
  tile_r2(r2, xi, xj, Xs, cd.D, i0, j0, t, tx, ty);
  const double sf2 = sp[SP_SF2], rqa = sp[SP_RQA], inv_ks = 1.0 / sp[SP_KSCALE];
  const double sfs = sf2 * inv_ks;  // K / (sn2_div * sn2_mult), :2416
  ExpC ex;
  ex.load();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
      double v;
      if (i < n && j < n) {
        v = pair_eval_t<KIND, DEG>(r2[a][c], sfs, rqa, ex).K;
        if (i == j) v += dvec[i];
      } else {
        v = (i == j) ? 1.0 : 0.0;
      }
      A[(size_t)i * lda + j] = (T)v;
    }
}

template <typename T, int KIND, int DEG>
__device__ __forceinline__ void build_tile(const CovDesc& cd, const double* __restrict__ Xs_all,
                                           const double* __restrict__ sp_all, const double* __restrict__ dvec_all,
                                           int n, int npad, T* __restrict__ A_all, long long sA, int lda, int tile,
                                           int b, double (*xi)[DCH + 1], double (*xj)[DCH + 1]) {
  build_tile_at<T, KIND, DEG>(cd, Xs_all, sp_all, dvec_all, n, npad, A_all, sA, lda, tile, b, b, xi, xj);
}

// grid = (number of lower tiles to build, batch); tile0 = index of the first one
template <typename T, int KIND, int DEG>
__global__ __launch_bounds__(256) void build_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                    const double* __restrict__ sp_all,
                                                    const double* __restrict__ dvec_all, int n,
                                                    int npad, T* __restrict__ A_all, long long sA,
                                                    int lda, int tile0) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  build_tile<T, KIND, DEG>(cd, Xs_all, sp_all, dvec_all, n, npad, A_all, sA, lda, tile0 + blockIdx.x, blockIdx.y, xi,
                           xj);
}

// The tail of the build (tiles tile0 .. ntiles-1 of every sample) as a persistent launch that stays off the
// same reserved CUs as a deferred GEMM (common.h: cu_reserve_bail; blocks that land on one return at once): it runs on a side
// stream UNDER the factorization of the first rows, whose leaves need whole empty CUs.  ctr: zeroed counter.
template <typename T, int KIND, int DEG>
__global__ __launch_bounds__(256) void build_persist_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                            const double* __restrict__ sp_all,
                                                            const double* __restrict__ dvec_all, int n, int npad,
                                                            T* __restrict__ A_all, long long sA, int lda, int tile0,
                                                            int ntiles, int batch,
                                                            const unsigned short* __restrict__ reserve, int* ctr) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  __shared__ int next;
  // ctr: a zeroed group of CTR_STRIDE counters; [0] is the tile queue
  if (reserve && cu_reserve_bail(reserve, ctr)) return;
  const int total = (ntiles - tile0) * batch;
  for (;;) {
    if (threadIdx.x == 0) next = atomicAdd(ctr, 1);
    __syncthreads();
    const int idx = __builtin_amdgcn_readfirstlane(next);
    __syncthreads();  // everyone holds idx before thread 0 overwrites it; also fences xi / xj between tiles
    if (idx >= total) break;
    build_tile<T, KIND, DEG>(cd, Xs_all, sp_all, dvec_all, n, npad, A_all, sA, lda, tile0 + idx / batch, idx % batch,
                             xi, xj);
  }
}

// ---------------------------------------------------------------------------------
// Gradient contraction over the lower triangle of Q = Kinv/sl - alpha alpha^T:
//   part[b][tile][p], p < P = cov_N + 1:  sum_ij w_ij Q_ij dK_ij/dtheta_p  (w = 2 off-diag)
//                                        last slot: trace(Q)
//   diagQ[b][i] = Q_ii
// grid = (lower tiles of npad/64, batch).  dK is recomputed from the scaled inputs.
// ---------------------------------------------------------------------------------
#ifndef GPC_TRACE_WPS
#define GPC_TRACE_WPS 1
#endif
template <typename T, int KIND, int DEG>
__global__ __launch_bounds__(256, GPC_TRACE_WPS) void trace_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                    const double* __restrict__ sp_all,
                                                    const double* __restrict__ alpha_all, int n,
                                                    int npad, const T* __restrict__ Kinv_all,
                                                    long long sK, int ldk,
                                                    double* __restrict__ part_all, int ntiles,
                                                    double* __restrict__ diagQ_all) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  extern __shared__ double wpart[];  // [4][P4], P4 = P rounded up to a multiple of 4 (+4)
  if constexpr (sizeof(T) == 4) {
    trace_tile32<KIND, DEG>(cd, Xs_all, sp_all, alpha_all, n, npad, Kinv_all, sK, ldk, part_all, ntiles, diagQ_all,
                            reinterpret_cast<float*>(&xi[0][0]), reinterpret_cast<float*>(&xj[0][0]), wpart);
    return;
  }
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, b = blockIdx.y;
  const int lane = t & 63, w = t >> 6;
  const int P = cd.cov_N + 1;
  const int P4 = ((P + 3) & ~3) + 4;
  constexpr bool ISO = (KIND == K_SE_ISO || KIND == K_MATERN_ISO);
  int ti, tj;
  lower_tile(blockIdx.x, ti, tj);
  const int i0 = ti * CT, j0 = tj * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* sp = sp_all + (size_t)b * SP_STRIDE;
  const double* alpha = alpha_all + (size_t)b * npad;
  const T* Kinv = Kinv_all + (size_t)b * sK;
  double* part = part_all + ((size_t)b * ntiles + blockIdx.x) * P;

  // the 16 entries of K^-1 and the alpha values of this thread first: their latency hides under the
  // distance sweep
#ifndef GPC_TRACE_NOPREFETCH
  T kin[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) kin[a][c] = Kinv[(size_t)(i0 + ty + 16 * a) * ldk + j0 + tx + 16 * c];
#endif
  double r2[4][4];
  tile_r2(r2, xi, xj, Xs, cd.D, i0, j0, t, tx, ty);
  double al_i[4], al_j[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    al_i[a] = alpha[i0 + ty + 16 * a];
    al_j[a] = alpha[j0 + tx + 16 * a];
  }

  const double sf2 = sp[SP_SF2], rqa = sp[SP_RQA], invsl = 1.0 / sp[SP_SL];
  ExpC ex;
  ex.load();
  double qF[4][4];
  double g_sf = 0.0, g_a = 0.0, g_iso = 0.0, trq = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
      const bool valid = (i < n) && (j <= i);
      double qf = 0.0;
      if (valid) {
#ifndef GPC_TRACE_NOPREFETCH
        const double kv = (double)kin[a][c];
#else
        const double kv = (double)Kinv[(size_t)i * ldk + j];
#endif
        const double Q = fma(kv, invsl, -al_i[a] * al_j[c]);
        const double qw = (i == j) ? Q : 2.0 * Q;
        const PairVal pv = pair_eval_t<KIND, DEG>(r2[a][c], sf2, rqa, ex);
        g_sf = fma(qw, 2.0 * pv.K, g_sf);
        if constexpr (KIND == K_RQ) g_a = fma(qw, pv.Ka, g_a);
        qf = qw * pv.F;
        if constexpr (ISO) g_iso = fma(qf, r2[a][c], g_iso);
        if (i == j) {
          trq += Q;
          diagQ_all[(size_t)b * npad + i] = Q;
        }
      }
      qF[a][c] = qf;  // masked entries: exactly 0 (their d^2 is finite, so 0 * d^2 = 0); a valid
                      // Matern-1 diagonal entry is +-inf and inf * 0 = NaN, as in the reference
    }

  const int own = lane >> 4;  // which of the four values of a wave_sum4 this lane ends up holding
  if constexpr (ISO) {
    const double u = wave_sum4(g_iso, g_sf, trq, 0.0, lane);
    if ((lane & 15) == 0 && own < 3) wpart[w * P4 + own] = u;  // slots 0, 1, 2 = P - 1
  } else {
    // second sweep over the input dimensions, four at a time: G_h = sum_e qF[e] * d_h[e]^2
    for (int h0 = 0; h0 < cd.D; h0 += DCH) {
      const int dc = min(DCH, cd.D - h0);
      if (cd.D > DCH) {  // restage (single pass when D <= 32: LDS still holds it)
        __syncthreads();
        stage_x(xi, Xs, cd.D, i0, h0, dc, t);
        stage_x(xj, Xs, cd.D, j0, h0, dc, t);
        __syncthreads();
      }
      for (int h = 0; h < dc; h += 4) {
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int hh = h + q;
          if (hh >= dc) break;  // block-uniform: the surplus slots of the last group stay 0 and are not stored
          double vi[4], vj[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) vi[a] = xi[ty + 16 * a][hh];
#pragma unroll
          for (int c = 0; c < 4; ++c) vj[c] = xj[tx + 16 * c][hh];
          double s = 0.0;
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const double d = vi[a] - vj[c];
              s = fma(qF[a][c] * d, d, s);
            }
          s4[q] = s;
        }
        const double u = wave_sum4(s4[0], s4[1], s4[2], s4[3], lane);
        if ((lane & 15) == 0 && h + own < dc) wpart[w * P4 + h0 + h + own] = u;
      }
    }
    const double u = wave_sum4(g_sf, g_a, trq, 0.0, lane);
    if ((lane & 15) == 0) {
      if (own == 0) wpart[w * P4 + cd.D] = u;
      if (own == 1 && KIND == K_RQ) wpart[w * P4 + cd.D + 1] = u;
      if (own == 2) wpart[w * P4 + P - 1] = u;
    }
  }
  __syncthreads();
  for (int p = t; p < P; p += 256)
    part[p] = wpart[p] + wpart[P4 + p] + wpart[2 * P4 + p] + wpart[3 * P4 + p];
}

// host-side dispatch on (kernel family, degree): one template instance per combination
#define GPC_COV_DISPATCH(KERNEL, T, cd, grid, block, shm, st, ...)                                            \
  do {                                                                                                      \
    switch ((cd).kind * 8 + ((cd).kind == K_MATERN || (cd).kind == K_MATERN_ISO ? (cd).degree : 0)) {      \
      case K_SE * 8: hipLaunchKernelGGL((KERNEL<T, K_SE, 0>), grid, block, shm, st, __VA_ARGS__); break;    \
      case K_RQ * 8: hipLaunchKernelGGL((KERNEL<T, K_RQ, 0>), grid, block, shm, st, __VA_ARGS__); break;    \
      case K_SE_ISO * 8: hipLaunchKernelGGL((KERNEL<T, K_SE_ISO, 0>), grid, block, shm, st, __VA_ARGS__); break; \
      case K_MATERN * 8 + 1: hipLaunchKernelGGL((KERNEL<T, K_MATERN, 1>), grid, block, shm, st, __VA_ARGS__); break; \
      case K_MATERN * 8 + 3: hipLaunchKernelGGL((KERNEL<T, K_MATERN, 3>), grid, block, shm, st, __VA_ARGS__); break; \
      case K_MATERN * 8 + 5: hipLaunchKernelGGL((KERNEL<T, K_MATERN, 5>), grid, block, shm, st, __VA_ARGS__); break; \
      case K_MATERN_ISO * 8 + 1: hipLaunchKernelGGL((KERNEL<T, K_MATERN_ISO, 1>), grid, block, shm, st, __VA_ARGS__); break; \
      case K_MATERN_ISO * 8 + 3: hipLaunchKernelGGL((KERNEL<T, K_MATERN_ISO, 3>), grid, block, shm, st, __VA_ARGS__); break; \
      case K_MATERN_ISO * 8 + 5: hipLaunchKernelGGL((KERNEL<T, K_MATERN_ISO, 5>), grid, block, shm, st, __VA_ARGS__); break; \
      default: break;                                                                                       \
    }                                                                                                       \
  } while (0)

// out[b][p] = sum_tile part[b][tile][p]  in a fixed order.  grid = (P, batch)
__global__ __launch_bounds__(256) void reduce_parts_kernel(const double* __restrict__ part, int ntiles,
                                                           int P, double* __restrict__ out) {
  __shared__ double sh4[4];
  const int p = blockIdx.x, b = blockIdx.y;
  double s = 0.0;
  for (int i = threadIdx.x; i < ntiles; i += 256) s += part[((size_t)b * ntiles + i) * P + p];
  s = block_sum_256(s, sh4);
  if (threadIdx.x == 0) out[(size_t)b * P + p] = s;
}

// ---------------------------------------------------------------------------------
// predict (round 4): cross covariance Ks[b] (npad x mpad, zero padding) = K(Xs, Xss) in 64 x 64 tiles -- inputs staged
// in LDS, the pair functor of the kernel build (the same K values as the training matrix, to the bit, where a query
// point is a training point) -- with the mean product fused in: mupart[b][ti][j] = sum over the 64 rows of tile row ti
// of Ks[i][j] alpha[i], summed over the tile rows afterwards (colpart_reduce_kernel), so mu = Ks^T alpha
// (gaussian_process.py:1747) never re-reads Ks.  grid = (mpad/64, npad/64, batch)
// ---------------------------------------------------------------------------------
template <typename T, int KIND, int DEG>
__global__ __launch_bounds__(256) void cross_tile_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                         const double* __restrict__ Xss_all,
                                                         const double* __restrict__ sp_all,
                                                         const double* __restrict__ alpha_all, int astride, int n,
                                                         int npad, int m, int mpad, T* __restrict__ Ks_all, long long sKs,
                                                         double* __restrict__ mupart_all) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  __shared__ double red[4][CT];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, b = blockIdx.z;
  const int lane = t & 63, w = t >> 6;
  const int i0 = blockIdx.y * CT, j0 = blockIdx.x * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* Xss = Xss_all + (size_t)b * mpad * cd.D;
  const double* sp = sp_all + (size_t)b * SP_STRIDE;
  const double* alpha = alpha_all + (size_t)b * astride;
  T* Ks = Ks_all + (size_t)b * sKs;
  double r2[4][4];
  tile_r2_ab(r2, xi, xj, Xs, Xss, cd.D, i0, j0, t, tx, ty);
  const double sf2 = sp[SP_SF2], rqa = sp[SP_RQA];
  ExpC ex;
  ex.load();
  double al[4], s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int a = 0; a < 4; ++a) al[a] = (i0 + ty + 16 * a) < n ? alpha[i0 + ty + 16 * a] : 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
      double v = 0.0;
      if (i < n && j < m) v = pair_eval_t<KIND, DEG>(r2[a][c], sf2, rqa, ex).K;
      const T vt = (T)v;
      Ks[(size_t)i * mpad + j] = vt;
      s[c] = fma((double)vt, al[a], s[c]);  // the stored value, so that mu is the product with what the solves read
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    s[c] += __shfl_xor(s[c], 16, 64);
    s[c] += __shfl_xor(s[c], 32, 64);
  }
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[w][tx + 16 * c] = s[c];
  }
  __syncthreads();
  if (t < CT)
    mupart_all[((size_t)b * (npad / CT) + blockIdx.y) * mpad + j0 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// out[b][j] = sum_t part[b][t][j], t ascending.  grid = (mpad/256, batch)
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const double* __restrict__ part, int nt, int mpad,
                                                             double* __restrict__ out) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= mpad) return;
  double s = 0.0;
  for (int k = 0; k < nt; ++k) s += part[((size_t)b * nt + k) * mpad + j];
  out[(size_t)b * mpad + j] = s;
}

// ---------------------------------------------------------------------------------
// Cross covariance Ks[b] (npad x mpad, row-major, zero padding) = K(Xs, Xss).
// One thread per entry; both operands are small and L2 resident.
// grid = (mpad/64, npad/4, batch), block = (64, 4)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void cross_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                             const double* __restrict__ Xss_all, const double* __restrict__ sp_all,
                             int n, int npad, int m, int mpad, T* __restrict__ Ks_all, long long sKs) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= npad || j >= mpad) return;
  double v = 0.0;
  if (i < n && j < m) {
    const double* xi = Xs_all + ((size_t)b * npad + i) * cd.D;
    const double* xj = Xss_all + ((size_t)b * mpad + j) * cd.D;
    double r2 = 0.0;
    for (int h = 0; h < cd.D; ++h) {
      const double d = xi[h] - xj[h];
      r2 += d * d;
    }
    const double* sp = sp_all + (size_t)b * SP_STRIDE;
    v = pair_eval(cd.kind, cd.degree, r2, sp[SP_SF2], sp[SP_RQA]).K;
  }
  Ks_all[(size_t)b * sKs + (size_t)i * mpad + j] = (T)v;
}

// ---------------------------------------------------------------------------------
// ks[b][i] = k(x_i, x_n) for i < n, 0 for i >= n, from the scaled inputs Xs[b] (row n = the
// appended point).  grid = (npad/256, batch)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_vec_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                        const double* __restrict__ sp_all, int n, int npad,
                                                        double* __restrict__ ks_all) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npad) return;
  double v = 0.0;
  if (i < n) {
    const double* xi = Xs_all + ((size_t)b * npad + i) * cd.D;
    const double* xn = Xs_all + ((size_t)b * npad + n) * cd.D;
    double r2 = 0.0;
    for (int h = 0; h < cd.D; ++h) {
      const double d = xi[h] - xn[h];
      r2 += d * d;
    }
    const double* sp = sp_all + (size_t)b * SP_STRIDE;
    v = pair_eval(cd.kind, cd.degree, r2, sp[SP_SF2], sp[SP_RQA]).K;
  }
  ks_all[(size_t)b * npad + i] = v;
}

// ---------------------------------------------------------------------------------
// Bayesian-quadrature kernel means (gaussian_process.py:1908-1921), SE kernel only:
//   z[b][n][j] = exp( ln sf2 + sum ln ell - sum_l ln tau_jl - 1/2 sum_l ((mu_jl - X_nl)/tau_jl)^2 )
//   tau_jl = sqrt(sigma_jl^2 + ell_l^2),  ell_l = dv[b][l] (the SE scaling divides X by ell)
// laid out like a cross covariance (npad x mpad, zero padding).
// grid = (mpad/64, npad/4, batch), block = (64, 4)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void quad_z_kernel(const double* __restrict__ X, const double* __restrict__ mu,
                              const double* __restrict__ sigma, const double* __restrict__ mul_all,
                              const double* __restrict__ dv_all, const double* __restrict__ sp_all, int n,
                              int npad, int m, int mpad, int D, T* __restrict__ Z_all, long long sZ) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= npad || j >= mpad) return;
  double v = 0.0;
  if (i < n && j < m) {
    const double* ell = dv_all + (size_t)b * D;
    const double sf2 = sp_all[(size_t)b * SP_STRIDE + SP_SF2];
    double lnnf = log(sf2), acc = 0.0;
    for (int l = 0; l < D; ++l) {
      const double sg = sigma[(size_t)j * D + l];
      const double tau = sqrt(sg * sg + ell[l] * ell[l]);
      lnnf += log(ell[l]) - log(tau);
      const double d = (mu[(size_t)j * D + l] - X[(size_t)i * D + l]) / tau;
      acc += d * d;
    }
    v = exp(lnnf - 0.5 * acc);
  }
  Z_all[(size_t)b * sZ + (size_t)i * mpad + j] = (T)v;
}

// ---------------------------------------------------------------------------------
// Standalone compute(): K (n x m) and optionally dK (n x n x cov_N) in double.
// Xa: scaled rows (n x D); Xb: scaled cols (m x D).  grid = (ceil(m/64), ceil(n/4))
// ---------------------------------------------------------------------------------
__global__ void full_cov_kernel(CovDesc cd, const double* __restrict__ Xa,
                                const double* __restrict__ Xb, double sf2, double rqa, int n, int m,
                                double* __restrict__ K, double* __restrict__ dK) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (i >= n || j >= m) return;
  const double* xi = Xa + (size_t)i * cd.D;
  const double* xj = Xb + (size_t)j * cd.D;
  double r2 = 0.0;
  for (int h = 0; h < cd.D; ++h) {
    const double d = xi[h] - xj[h];
    r2 += d * d;
  }
  const PairVal pv = pair_eval(cd.kind, cd.degree, r2, sf2, rqa);
  K[(size_t)i * m + j] = pv.K;
  if (dK) {
    double* g = dK + ((size_t)i * m + j) * cd.cov_N;
    if (cov_is_iso(cd.kind)) {
      g[0] = pv.F * r2;
      g[1] = 2.0 * pv.K;
    } else {
      for (int h = 0; h < cd.D; ++h) {
        const double d = xi[h] - xj[h];
        g[h] = pv.F * (d * d);
      }
      g[cd.D] = 2.0 * pv.K;
      if (cd.kind == K_RQ) g[cd.D + 1] = pv.Ka;
    }
  }
}

}  // namespace gpc

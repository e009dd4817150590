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
  for (int e = t; e < CT * dc; e += 256) {
    const int r = e / dc, h = e % dc;
    sh[r][h] = Xs[(size_t)(r0 + r) * D + h0 + h];
  }
}

// ---------------------------------------------------------------------------------
// A[b] (lower 64x64 tiles) = K(Xs, Xs) / sp[SP_KSCALE] + diag(dvec), identity in the padding.
// grid = (lower tiles of npad/64, batch)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void build_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                    const double* __restrict__ sp_all,
                                                    const double* __restrict__ dvec_all, int n,
                                                    int npad, T* __restrict__ A_all, long long sA,
                                                    int lda) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, b = blockIdx.y;
  int ti, tj;
  {
    int i = (int)((sqrtf(8.f * (float)blockIdx.x + 1.f) - 1.f) * 0.5f);
    while (i * (i + 1) / 2 > (int)blockIdx.x) --i;
    while ((i + 1) * (i + 2) / 2 <= (int)blockIdx.x) ++i;
    ti = i;
    tj = blockIdx.x - i * (i + 1) / 2;
  }
  const int i0 = ti * CT, j0 = tj * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* sp = sp_all + (size_t)b * SP_STRIDE;
  const double* dvec = dvec_all + (size_t)b * npad;
  T* A = A_all + (size_t)b * sA;

  double r2[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) r2[a][c] = 0.0;

  for (int h0 = 0; h0 < cd.D; h0 += DCH) {
    const int dc = min(DCH, cd.D - h0);
    __syncthreads();
    stage_x(xi, Xs, cd.D, i0, h0, dc, t);
    stage_x(xj, Xs, cd.D, j0, h0, dc, t);
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
          r2[a][c] += d * d;
        }
    }
  }
  const double sf2 = sp[SP_SF2], rqa = sp[SP_RQA], inv_ks = 1.0 / sp[SP_KSCALE];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * c;
      double v;
      if (i < n && j < n) {
        v = pair_eval(cd.kind, cd.degree, r2[a][c], sf2, rqa).K * inv_ks;  // K / (sn2_div * sn2_mult), :2416
        if (i == j) v += dvec[i];
      } else {
        v = (i == j) ? 1.0 : 0.0;
      }
      A[(size_t)i * lda + j] = (T)v;
    }
}

// ---------------------------------------------------------------------------------
// Gradient contraction over the lower triangle of Q = Kinv/sl - alpha alpha^T:
//   part[b][tile][p], p < P = cov_N + 1:  sum_ij w_ij Q_ij dK_ij/dtheta_p  (w = 2 off-diag)
//                                        last slot: trace(Q)
//   diagQ[b][i] = Q_ii
// grid = (lower tiles of npad/64, batch).  dK is recomputed from the scaled inputs.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void trace_kernel(CovDesc cd, const double* __restrict__ Xs_all,
                                                    const double* __restrict__ sp_all,
                                                    const double* __restrict__ alpha_all, int n,
                                                    int npad, const T* __restrict__ Kinv_all,
                                                    long long sK, int ldk,
                                                    double* __restrict__ part_all, int ntiles,
                                                    double* __restrict__ diagQ_all) {
  __shared__ double xi[CT][DCH + 1];
  __shared__ double xj[CT][DCH + 1];
  extern __shared__ double wpart[];  // [4][P]
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4, b = blockIdx.y;
  const int lane = t & 63, w = t >> 6;
  const int P = cd.cov_N + 1;
  int ti, tj;
  {
    int i = (int)((sqrtf(8.f * (float)blockIdx.x + 1.f) - 1.f) * 0.5f);
    while (i * (i + 1) / 2 > (int)blockIdx.x) --i;
    while ((i + 1) * (i + 2) / 2 <= (int)blockIdx.x) ++i;
    ti = i;
    tj = blockIdx.x - i * (i + 1) / 2;
  }
  const int i0 = ti * CT, j0 = tj * CT;
  const double* Xs = Xs_all + (size_t)b * npad * cd.D;
  const double* sp = sp_all + (size_t)b * SP_STRIDE;
  const double* alpha = alpha_all + (size_t)b * npad;
  const T* Kinv = Kinv_all + (size_t)b * sK;
  double* part = part_all + ((size_t)b * ntiles + blockIdx.x) * P;
  const bool iso = cov_is_iso(cd.kind);

  double r2[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) r2[a][c] = 0.0;
  for (int h0 = 0; h0 < cd.D; h0 += DCH) {
    const int dc = min(DCH, cd.D - h0);
    __syncthreads();
    stage_x(xi, Xs, cd.D, i0, h0, dc, t);
    stage_x(xj, Xs, cd.D, j0, h0, dc, t);
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
          r2[a][c] += d * d;
        }
    }
  }

  const double sf2 = sp[SP_SF2], rqa = sp[SP_RQA], invsl = 1.0 / sp[SP_SL];
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
        const double Q = (double)Kinv[(size_t)i * ldk + j] * invsl - alpha[i] * alpha[j];
        const double qw = (i == j) ? Q : 2.0 * Q;
        const PairVal pv = pair_eval(cd.kind, cd.degree, r2[a][c], sf2, rqa);
        g_sf += qw * (2.0 * pv.K);
        g_a += qw * pv.Ka;
        qf = qw * pv.F;
        if (iso) g_iso += qf * r2[a][c];
        if (i == j) {
          trq += Q;
          diagQ_all[(size_t)b * npad + i] = Q;
        }
      }
      qF[a][c] = qf;
    }

  // per-wave sums into wpart[w][p]
  auto put = [&](int p, double v) {
    v = wave_sum(v);
    if (lane == 0) wpart[w * P + p] = v;
  };
  if (iso) {
    put(0, g_iso);
    put(1, g_sf);
  } else {
    // second sweep over the input dimensions: G_h = sum_e qF[e] * d_h[e]
    for (int h0 = 0; h0 < cd.D; h0 += DCH) {
      const int dc = min(DCH, cd.D - h0);
      if (cd.D > DCH) {  // restage (single pass when D <= 32: LDS still holds it)
        __syncthreads();
        stage_x(xi, Xs, cd.D, i0, h0, dc, t);
        stage_x(xj, Xs, cd.D, j0, h0, dc, t);
        __syncthreads();
      }
      for (int h = 0; h < dc; ++h) {
        double vi[4], vj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) vi[a] = xi[ty + 16 * a][h];
#pragma unroll
        for (int c = 0; c < 4; ++c) vj[c] = xj[tx + 16 * c][h];
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double d = vi[a] - vj[c];
            // select, not multiply: a masked entry must not turn 0 * inf into NaN,
            // a valid Matern-1 diagonal entry must (reference semantics).
            s += (qF[a][c] != 0.0 || qF[a][c] != qF[a][c]) ? qF[a][c] * (d * d) : 0.0;
          }
        put(h0 + h, s);
      }
    }
    put(cd.D, g_sf);
    if (cd.kind == K_RQ) put(cd.D + 1, g_a);
  }
  put(P - 1, trq);
  __syncthreads();
  for (int p = t; p < P; p += 256)
    part[p] = wpart[p] + wpart[P + p] + wpart[2 * P + p] + wpart[3 * P + p];
}

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

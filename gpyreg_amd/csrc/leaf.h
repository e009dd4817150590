// leaf.h -- 128 x 128 Cholesky factor AND its inverse in one workgroup (batched).
//
// This is the "diagonal potrf + trsm seed" of the blocked factorization: for the
// diagonal block D it produces L (D = L L^T, written to the lower triangle of the
// block in A) and W = L^-1 (full 128 x 128 tile written to the W buffer, zeros above
// the diagonal), plus sum(log diag L) and the LAPACK-style info flag.
//
// leaf3 is MFMA-blocked (16-wide panels, the block held as accumulator tiles).  The scalar elimination it
// replaced (84 us per leaf against 46 us; two pivots per barrier) was removed in round 2.
#pragma once
#include "common.h"

#include <type_traits>

namespace gpc {

// 1/sqrt(x) to working precision from the hardware estimate plus Newton steps (the
// pivot's square root and reciprocal sit on the serial critical path of every step)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}
__device__ __forceinline__ float fast_rsqrt(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  r = r * (1.5f - 0.5f * x * r * r);
  return r;
}

// ---- leaf3: MFMA-blocked 128 x 128 Cholesky + inverse ----------------------------------------
// Right-looking, panel width 16 (= one MFMA tile), 8 panels.  The block lives in registers as
// MFMA accumulator tiles: wave w owns block rows w and 7-w (9 lower tiles each), slot (i, j) holds
// A_ij until panel j is eliminated and V_ij afterwards, where V is the identity carried through
// the same elimination (V = L^-1 at the end) -- the in-place trick of the scalar leaf at tile
// granularity.  Per panel P (two block barriers):
//   publish   column P of the slots (A_iP, i >= P) and row P (V_Pj, j < P) to LDS        | barrier
//   diag      every wave factors A_PP = L_PP L_PP^T and inverts it (W_PP) in registers, one
//             matrix row per lane, pivots / multipliers broadcast with v_readlane: the only
//             serial part, 16 dependent pivots, no LDS and no barrier inside
//   solve     L_iP = A_iP W_PP^T (own rows i > P; final, written out);
//             row owner: W_Pj = W_PP V_Pj (final row P of the inverse), W_PP                | barrier
//   update    own rows i > P:  slot(i,j) -= L_iP L_jP^T (j > P)   |   -= L_iP W_Pj (j <= P)
// All products are v_mfma 16x16x4 on 16 x 16 tiles read from LDS images ([row][17] / [k][144],
// conflict free).  224 tile products per leaf instead of 128 barrier-separated rank-1 steps.
// Cross-lane traffic of the 16 x 16 diagonal factorization: DPP row_newbcast (gfx90a+) puts lane N
// of each row of 16 lanes into every lane of that row with one full-rate VALU move (64-bit form
// for fp64), results stay in vector registers.  The moves and the two FMAs they feed are emitted
// as one asm block per (pivot, row): left to the compiler the two FMA chains are scheduled
// apart and every broadcast value is parked in AGPRs / scratch in between (measured: 84 us per
// leaf instead of 45).  `s_nop 1` covers the VALU-write -> DPP-read hazard, which the hazard
// recognizer cannot see inside inline asm.
template <int N>
__device__ __forceinline__ double bcast16(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
  return r;
}
template <int N>
__device__ __forceinline__ float bcast16(float v) {
  float r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
  return r;
}
// dj -= dk * bcast_J(dk);  wj -= bcast_J(dk) * wk
template <int J>
__device__ __forceinline__ void elim16(double& dj, double& wj, double dk, double wk) {
  double t;
  asm volatile(
      "s_nop 1\n\tv_mov_b64_dpp %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fma_f64 %0, -%3, %2, %0\n\tv_fma_f64 %1, -%2, %4, %1"
      : "+v"(dj), "+v"(wj), "=&v"(t)
      : "v"(dk), "v"(wk), "n"(J));
}
template <int J>
__device__ __forceinline__ void elim16(float& dj, float& wj, float dk, float wk) {
  float t;
  asm volatile(
      "s_nop 1\n\tv_mov_b32_dpp %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fma_f32 %0, -%3, %2, %0\n\tv_fma_f32 %1, -%2, %4, %1"
      : "+v"(dj), "+v"(wj), "=&v"(t)
      : "v"(dk), "v"(wk), "n"(J));
}
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// d: lane l holds row (l & 15) of an SPD 16 x 16 block (entries right of the diagonal are not
// used; the four rows of 16 lanes carry identical copies).  On return d holds the same row of
// its Cholesky factor L and wv the COLUMN (l & 15) of W = L^-1 (wv[i] = W[i][l & 15]).
// badk: first non-positive pivot, or -1.
template <typename T>
__device__ __forceinline__ void diag16(T (&d)[16], T (&wv)[16], int l15, int& badk) {
  // The forward substitution L W = I (lane c solves column c of W; entries above the diagonal
  // come out as exact zeros) rides on the elimination: the multiplier L_jk broadcast for the
  // Schur update of pivot k is the one W needs, so every broadcast feeds two FMAs.
#pragma unroll
  for (int i = 0; i < 16; ++i) wv[i] = (l15 == i) ? (T)1 : (T)0;
  static_for<0, 16>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const T s = bcast16<k>(d[k]);
    if (!(s > (T)0) && badk < 0) badk = k;
    const T r = fast_rsqrt(s);
    d[k] = d[k] * r;    // column k of L (lane k: L_kk = s / sqrt(s))
    wv[k] = wv[k] * r;  // W[k][c] = (delta_kc - sum_{m<k} L_km W[m][c]) / L_kk
    static_for<k + 1, 16>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      elim16<j>(d[j], wv[j], d[k], wv[k]);
    });
  });
}

namespace leaf3 {
constexpr int LDC = 17;   // [row][k] images
constexpr int LDR = 144;  // [k][col] images
template <typename A, typename T>
__device__ __forceinline__ void mma_slot(A& c, T a, T b) {
  c = MM<T>::mma(a, b, c);
}
}  // namespace leaf3

template <typename T, int P>
__device__ __forceinline__ void leaf3_panel(typename MM<T>::acc_t (&S)[2][8], T* __restrict__ colbuf,
                                            T* __restrict__ lcol, T* __restrict__ rowbuf,
                                            T* __restrict__ urow, T* __restrict__ myW,
                                            T* __restrict__ Ab, int lda, int w, int lane, double& lg,
                                            int& bad) {
  using acc_t = typename MM<T>::acc_t;
  using namespace leaf3;
  const int l15 = lane & 15, lq = lane >> 4;
  // 1. publish
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r ? 7 - w : w;
    if (i >= P) {
#pragma unroll
      for (int e = 0; e < 4; ++e) colbuf[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = S[r][P][e];
    }
    if (i == P) {
#pragma unroll
      for (int j = 0; j < P; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) rowbuf[MM<T>::row_of(lane, e) * LDR + 16 * j + l15] = S[r][j][e];
    }
  }
  __syncthreads();
  // 2. diagonal block, redundantly in every wave
  T d[16], wv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) d[c] = colbuf[(16 * P + l15) * LDC + c];
  int badk = -1;
#if defined(GPC_LEAF3_NODIAG)  // timing experiment only: results are wrong
#pragma unroll
  for (int i = 0; i < 16; ++i) wv[i] = (l15 == i) ? (T)1 : (T)0;
#else
  diag16<T>(d, wv, l15, badk);
#endif
  if (badk >= 0 && bad == 0) bad = 16 * P + badk + 1;
  if (lq == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) myW[i * LDC + l15] = wv[i];
  }
  if (w == 0 && lq == 0) {
    T dkk = d[0];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c <= l15) Ab[(size_t)(16 * P + l15) * lda + 16 * P + c] = d[c];
      if (c == l15) dkk = d[c];
    }
    lg += log((double)dkk);
  }
  // 3. panel solve (own rows below the panel) and the final row P of the inverse (its owner)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r ? 7 - w : w;
    if (i > P) {
      acc_t c = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        c = MM<T>::mma(colbuf[(16 * i + l15) * LDC + 4 * q + lq], myW[l15 * LDC + 4 * q + lq], c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * i + MM<T>::row_of(lane, e);
        Ab[(size_t)row * lda + 16 * P + l15] = c[e];
        lcol[row * LDC + l15] = c[e];
      }
    }
    if (i == P) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        acc_t c = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c = MM<T>::mma(myW[l15 * LDC + 4 * q + lq], rowbuf[(4 * q + lq) * LDR + 16 * j + l15], c);
        S[r][j] = c;
#pragma unroll
        for (int e = 0; e < 4; ++e) urow[MM<T>::row_of(lane, e) * LDR + 16 * j + l15] = c[e];
      }
      acc_t c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c[e] = myW[MM<T>::row_of(lane, e) * LDC + l15];
        urow[MM<T>::row_of(lane, e) * LDR + 16 * P + l15] = c[e];
      }
      S[r][P] = c;
    }
  }
  __syncthreads();
  // 4. trailing update of the own rows below the panel
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r ? 7 - w : w;
    if (i > P) {
      T a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = -lcol[(16 * i + l15) * LDC + 4 * q + lq];
      S[r][P] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j > i) continue;
        if (j > P) {
#pragma unroll
          for (int q = 0; q < 4; ++q) mma_slot(S[r][j], a[q], lcol[(16 * j + l15) * LDC + 4 * q + lq]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) mma_slot(S[r][j], a[q], urow[(4 * q + lq) * LDR + 16 * j + l15]);
        }
      }
    }
  }
}

// GPC_LEAF3_WPS = waves per SIMD the register allocation allows for: 2 caps the leaf at 256 registers so
// that a leaf block fits on a CU BESIDE one resident 128-tile GEMM block (240 VGPRs, 72 KB LDS).
#ifndef GPC_LEAF3_WPS
#define GPC_LEAF3_WPS 1
#endif
template <typename T>
__global__ __launch_bounds__(256, GPC_LEAF3_WPS) void leaf3_kernel(T* __restrict__ A, long long sA, int lda,
                                                    T* __restrict__ W, long long sW, int ldw, int off,
                                                    double* __restrict__ logdet, int* __restrict__ info,
                                                    int nvalid) {
  using acc_t = typename MM<T>::acc_t;
  using namespace leaf3;
  __shared__ T colbuf[TILE * LDC];
  __shared__ T lcol[TILE * LDC];
  __shared__ T rowbuf[16 * LDR];
  __shared__ T urow[16 * LDR];
  __shared__ T dgW[4][16 * LDC];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15;
  T* Ab = A + (size_t)blockIdx.x * sA;
  T* Wb = W + (size_t)blockIdx.x * sW;

  acc_t S[2][8];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r ? 7 - w : w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      S[r][j] = acc_t{0, 0, 0, 0};
      if (j <= i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * i + MM<T>::row_of(lane, e), col = 16 * j + l15;
          S[r][j][e] = (col <= row) ? Ab[(size_t)row * lda + col] : (T)0;
        }
      }
    }
  }
  double lg = 0.0;
  int bad = 0;
  // panels that start at or beyond nvalid are identity padding: L = I, W = I, already in place
  leaf3_panel<T, 0>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 16) leaf3_panel<T, 1>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 32) leaf3_panel<T, 2>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 48) leaf3_panel<T, 3>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 64) leaf3_panel<T, 4>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 80) leaf3_panel<T, 5>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 96) leaf3_panel<T, 6>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);
  if (nvalid > 112) leaf3_panel<T, 7>(S, colbuf, lcol, rowbuf, urow, dgW[w], Ab, lda, w, lane, lg, bad);

#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int i = r ? 7 - w : w;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * i + MM<T>::row_of(lane, e), col = 16 * j + l15;
        Wb[(size_t)row * ldw + col] = (j <= i) ? S[r][j][e] : (T)0;
      }
  }
  if (w == 0) {
    lg = wave_sum(lg);
    if (lane == 0) {
      if (bad) atomicCAS(info + blockIdx.x, 0, off + bad);
      atomicAdd(logdet + blockIdx.x, lg);
    }
  }
}

inline int g_leaf_version = 3;  // kept for the launch-graph key; one leaf kernel exists

}  // namespace gpc

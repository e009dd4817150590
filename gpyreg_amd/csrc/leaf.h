// leaf.h -- 128 x 128 Cholesky factor AND its inverse in one workgroup (batched).
//
// This is the "diagonal potrf + trsm seed" of the blocked factorization: for the
// diagonal block D it produces L (D = L L^T, written to the lower triangle of the
// block in A) and W = L^-1 (full 128 x 128 tile written to the W buffer, zeros above
// the diagonal), plus sum(log diag L) and the LAPACK-style info flag.
//
// Two kernels: leaf3 (default, further down) is MFMA-blocked; leaf2 (GPC_LEAF=2) is the scalar
// algorithm it replaced, kept as a cross-check: a right-looking elimination in which the
// 128 x 128 block lives in REGISTERS (each of the 256 threads owns an 8 x 8 set of entries,
// cyclically distributed: rows ty+16a, cols tx+16b) and only the pivot columns and pivot rows
// travel through LDS (two pivots per barrier).
// The same rank-1 update that eliminates column j of the Cholesky factor also
// advances the forward substitution L W = I, in place:
//     rows i > j :  M[i][k] -= c[i] * v[k]
//        c[i] = L[i][j]                      (pivot column / sqrt(pivot))
//        v[k] = L[k][j]   for k > j           -> Schur complement of the SPD block
//        v[k] = W[j][k]   for k <= j          -> W[i][k] accumulates -sum L[i][j] W[j][k]
// Columns k <= j of the register block no longer hold A (their L values have been
// written out) so they are reused for W: after 128 steps the registers hold W.
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

// =====================================================================================
// leaf2: in-register elimination, TWO pivots per barrier.
//
// Per pair of columns (j0, j1 = j0+1) the owners publish the two raw pivot columns and
// the two raw pivot rows; every thread redundantly factors the 2x2 pivot block
//     [p00  . ]        d0 = sqrt(p00), l10 = p10/d0, d1 = sqrt(p11 - l10^2)
//     [p10 p11]        inverse of [[d0,0],[l10,d1]] = [[r0,0],[w10,r1]], w10 = -l10 r0 r1
// and applies the rank-2 update  M[i][k] -= c0[i] v0[k] + c1[i] v1[k]  to rows i > j1:
//     c0 = C0 r0,  c1 = (C1 - c0 l10) r1                 (columns of L)
//     v0 = X0 r0,  v1 = (X1 - l10 v0) r1   with X = C (k > j1: Schur part)
//                                              or R (k < j0: rows of W)
//     inside the pair: v0[j0] = r0, v1[j0] = w10, v0[j1] = 0, v1[j1] = r1.
// L is staged in LDS and written out once, coalesced.
// =====================================================================================
constexpr int LDL = TILE + 1;  // LDS stride of the staged L tile

template <typename T, int JB>
__device__ __forceinline__ void leaf2_steps(T (&M)[8][8], T* __restrict__ pub, T* __restrict__ dbuf,
                                            T* __restrict__ Ls, int tx, int ty, int& bad, int nvalid) {
#pragma clang loop unroll(disable)
  for (int jj = 0; jj < 16; jj += 2) {
    const int j0 = JB * 16 + jj, j1 = j0 + 1;
    if (j0 >= nvalid) break;  // the rest of the tile is identity padding: nothing to eliminate
    T* C0 = pub + ((jj >> 1) & 1) * 4 * TILE;
    T* C1 = C0 + TILE;
    T* R0 = C1 + TILE;
    T* R1 = R0 + TILE;
    if (tx == jj) {
#pragma unroll
      for (int a = 0; a < 8; ++a) C0[ty + 16 * a] = M[a][JB];
    }
    if (tx == jj + 1) {
#pragma unroll
      for (int a = 0; a < 8; ++a) C1[ty + 16 * a] = M[a][JB];
    }
    if (ty == jj) {
#pragma unroll
      for (int b = 0; b < 8; ++b) R0[tx + 16 * b] = M[JB][b];
    }
    if (ty == jj + 1) {
#pragma unroll
      for (int b = 0; b < 8; ++b) R1[tx + 16 * b] = M[JB][b];
    }
    __syncthreads();
    const T p00 = C0[j0], p10 = C0[j1], p11 = C1[j1];
    const T r0 = fast_rsqrt(p00);
    const T d0 = p00 * r0;
    const T l10 = p10 * r0;
    const T s11 = fma(-l10, l10, p11);
    const T r1 = fast_rsqrt(s11);
    const T d1 = s11 * r1;
    const T w10 = -(l10 * r0) * r1;
    if (bad == 0) {
      if (!(p00 > (T)0))
        bad = j0 + 1;
      else if (!(s11 > (T)0))
        bad = j1 + 1;
    }
    if (tx == jj && ty == jj) {
      dbuf[j0] = d0;
      dbuf[j1] = d1;
    }

    T c0[8], c1[8], v0[8], v1[8];
#pragma unroll
    for (int a = JB; a < 8; ++a) {
      const int i = ty + 16 * a;
      c0[a] = C0[i] * r0;
      c1[a] = fma(-c0[a], l10, C1[i]) * r1;
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int k = tx + 16 * b;
      T x0, x1;
      if (b > JB) {
        x0 = C0[k];
        x1 = C1[k];
      } else if (b < JB) {
        x0 = R0[k];
        x1 = R1[k];
      } else {
        const bool right = tx > jj + 1;
        x0 = right ? C0[k] : R0[k];
        x1 = right ? C1[k] : R1[k];
      }
      T a0 = x0 * r0;
      T a1 = fma(-l10, a0, x1) * r1;
      if (b == JB) {
        if (tx == jj) {
          a0 = r0;
          a1 = w10;
        } else if (tx == jj + 1) {
          a0 = (T)0;
          a1 = r1;
        }
      }
      v0[b] = a0;
      v1[b] = a1;
    }
    // owners of the two pivot columns: stage L, recycle the registers for W
    if (tx == jj || tx == jj + 1) {
      const bool first = (tx == jj);
      const int jc = first ? j0 : j1;
#pragma unroll
      for (int a = JB; a < 8; ++a) {
        const int i = ty + 16 * a;
        if (i > j1) {
          Ls[i * LDL + jc] = first ? c0[a] : c1[a];
          M[a][JB] = (T)0;
        } else if (i == j1) {
          Ls[i * LDL + jc] = first ? l10 : d1;
        } else if (i == j0 && first) {
          Ls[i * LDL + jc] = d0;
        }
      }
    }
    // rank-2 update of every row below the pair
#pragma unroll
    for (int a = JB; a < 8; ++a) {
      const bool active = (a > JB) || (ty > jj + 1);
      if (active) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (b > JB && b > a) continue;  // strictly-upper blocks of the Schur part are never read
          M[a][b] = fma(-c1[a], v1[b], fma(-c0[a], v0[b], M[a][b]));
        }
      }
    }
    // the two pivot rows become rows j0, j1 of W
    if (ty == jj) {
#pragma unroll
      for (int b = 0; b < 8; ++b) M[JB][b] = (tx + 16 * b <= j0) ? v0[b] : (T)0;
    }
    if (ty == jj + 1) {
#pragma unroll
      for (int b = 0; b < 8; ++b) M[JB][b] = (tx + 16 * b <= j1) ? v1[b] : (T)0;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void leaf2_kernel(T* __restrict__ A, long long sA, int lda,
                                                    T* __restrict__ W, long long sW, int ldw, int off,
                                                    double* __restrict__ logdet, int* __restrict__ info,
                                                    int nvalid) {
  __shared__ T Ls[TILE * LDL];
  __shared__ T pub[8 * TILE];
  __shared__ T dbuf[TILE];
  __shared__ double red4[4];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  T* Ab = A + (size_t)blockIdx.x * sA;
  T* Wb = W + (size_t)blockIdx.x * sW;

  T M[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, k = tx + 16 * b;
      M[a][b] = (k <= i) ? Ab[(size_t)i * lda + k] : (T)0;
    }

  // columns >= nvalid (rounded up to a pivot pair) are identity padding: their L column is
  // e_j, their pivot is 1, and the registers already hold the matching rows/columns of W = I
  const int nelim = min(TILE, (nvalid + 1) & ~1);
  if (nelim < TILE) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int i = ty + 16 * a, k = tx + 16 * b;
        if (k >= nelim && k <= i) Ls[i * LDL + k] = (i == k) ? (T)1 : (T)0;
      }
    if (t >= nelim && t < TILE) dbuf[t] = (T)1;
  }
  int bad = 0;
  leaf2_steps<T, 0>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 16) leaf2_steps<T, 1>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 32) leaf2_steps<T, 2>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 48) leaf2_steps<T, 3>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 64) leaf2_steps<T, 4>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 80) leaf2_steps<T, 5>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 96) leaf2_steps<T, 6>(M, pub, dbuf, Ls, tx, ty, bad, nelim);
  if (nelim > 112) leaf2_steps<T, 7>(M, pub, dbuf, Ls, tx, ty, bad, nelim);

#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, k = tx + 16 * b;
      Wb[(size_t)i * ldw + k] = M[a][b];
    }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, k = tx + 16 * b;
      if (b <= a && k <= i) Ab[(size_t)i * lda + k] = Ls[i * LDL + k];
    }
  const double lg = block_sum_256(t < TILE ? log((double)dbuf[t]) : 0.0, red4);
  if (t == 0) {
    if (bad) atomicCAS(info + blockIdx.x, 0, off + bad);
    atomicAdd(logdet + blockIdx.x, lg);
  }
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

inline int g_leaf_version = 3;  // GPC_LEAF: 2 scalar elimination by pivot pairs, 3 MFMA-blocked (default)

}  // namespace gpc

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

// ---- leaf5: the same arithmetic as leaf3, software-pipelined over wave-specialised roles --------------
// leaf3 runs its phases one after the other on all four waves: publish | diag (16 dependent pivots, redundantly
// in every wave) | solve | update, two block barriers per panel.  The pivot chain is a third of its time and the
// other two thirds (LDS round trips, barriers, MFMA updates, global stores) sit serially between the chains.
// Here wave 0 ("diag wave", alone on its SIMD) does nothing but the chain: it factors and inverts A_PP the moment
// the tile is final, hands W_PP over through LDS and waits for A_(P+1)(P+1).  Waves 1..3 ("update waves") own
// the 36 accumulator slots (tile rows {7,2,1}, {6,3,0}, {5,4}) and do all the MFMA work of panel P while the diag
// wave is already inside the chain of panel P+1: the owner of tile row P+1 first computes L_(P+1)P and the one
// tile the next chain needs (A_(P+1)(P+1) -= L L^T, from its own data, before the panel barrier), signals, and
// only then joins the rest of the solve and the trailing update.  Hand-offs are LDS flags (release / acquire at
// workgroup scope); the one hardware barrier per panel separates "L_iP, W_Pj published" from their readers and the
// diag wave takes part in it at a moment when it has nothing to do.  Images read across that barrier are double
// buffered by panel parity.  Every tile product is the same MFMA sequence on the same operands as in leaf3, so the
// results are bit-identical (tests/test_gpu_kernels.py).
namespace leaf5 {
constexpr int LDC = 17;   // [row][k] images
constexpr int LDR = 144;  // [k][col] images
template <typename T>
struct Shared {
  T lcol[2][TILE * LDC];   // L_iP of panel P (parity P & 1), all tile rows
  T urow[2][16 * LDR];     // W_Pj, j <= P (final row P of the inverse)
  T colbuf[TILE * LDC];    // A_iP ahead of its solve; rows 16 i .. 16 i + 15 are touched by the owner of row i only
  T rowbuf[16 * LDR];      // V_Pj, j < P, ahead of the solve of panel P (written in update P-1, read by the owner of row P)
  T diagA[2][16 * LDC];    // A_PP for the diag wave
  T diagW[2][16 * LDC];    // W_PP from the diag wave
  int flagA;               // = P + 1 once A_PP is in diagA[P & 1]
  int flagW;               // = P + 1 once W_PP is in diagW[P & 1]
};
// tile rows of update wave u (0..2), slot r (0..2); -1: none
__device__ __forceinline__ int tile_row(int u, int r) { return r == 0 ? 7 - u : (r == 1 ? 2 + u : 1 - u); }

// Every wave reaches the end of every wait: a hand-off that never comes (a bug, not a data condition) is reported
// as a failed pivot after ~0.1 s instead of hanging the queue.
template <bool SLEEP>
__device__ __forceinline__ void wait_ge(int* f, int v, int& bad) {
  int n = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
    if (SLEEP) __builtin_amdgcn_s_sleep(1);
    if (++n > (1 << 21)) {
      bad = 1;
      break;
    }
  }
}
__device__ __forceinline__ void post(int* f, int v) {
  __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// block barrier that waits for the wave's LDS traffic only (global stores of L stay in flight)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// L_iP = A_iP W_PP^T for one owned tile row: colbuf image x wb fragments -> global L, lcol image
template <typename T>
__device__ __forceinline__ void solve_tile(const T* __restrict__ colbuf, const T (&wb)[4], T* __restrict__ lc,
                                           T* __restrict__ Ab, int lda, int i, int P, int lane) {
  using acc_t = typename MM<T>::acc_t;
  const int l15 = lane & 15, lq = lane >> 4;
  acc_t c = acc_t{0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) c = MM<T>::mma(colbuf[(16 * i + l15) * LDC + 4 * q + lq], wb[q], c);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = 16 * i + MM<T>::row_of(lane, e);
    Ab[(size_t)row * lda + 16 * P + l15] = c[e];
    lc[row * LDC + l15] = c[e];
  }
}

template <typename T, int P>
__device__ __forceinline__ void update_panel(typename MM<T>::acc_t (&S)[3][8], Shared<T>& sh, T* __restrict__ Ab,
                                             int lda, int u, int lane, int np, int& bad) {
  using acc_t = typename MM<T>::acc_t;
  const int l15 = lane & 15, lq = lane >> 4;
  T* lc = sh.lcol[P & 1];
  T* ur = sh.urow[P & 1];
  const T* dW = sh.diagW[P & 1];
  wait_ge<true>(&sh.flagW, P + 1, bad);
  T wb[4];  // B fragments of W_PP^T
#pragma unroll
  for (int q = 0; q < 4; ++q) wb[q] = dW[l15 * LDC + 4 * q + lq];
  // fast path: the owner of tile row P+1 hands the next diagonal tile to the diag wave first
  if constexpr (P + 1 < 8) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int i = tile_row(u, r);
      if (i == P + 1 && P + 1 < np) {
        solve_tile<T>(sh.colbuf, wb, lc, Ab, lda, i, P, lane);
        __builtin_amdgcn_wave_barrier();
        T a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          b[q] = lc[(16 * i + l15) * LDC + 4 * q + lq];
          a[q] = -b[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) S[r][P + 1] = MM<T>::mma(a[q], b[q], S[r][P + 1]);
        T* dA = sh.diagA[(P + 1) & 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) dA[MM<T>::row_of(lane, e) * LDC + l15] = S[r][P + 1][e];
        post(&sh.flagA, P + 2);
      }
    }
  }
  // the other owned tile rows below the panel; the owner of row P: final row P of the inverse
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int i = tile_row(u, r);
    if (i > P + 1 || (i == P + 1 && !(P + 1 < np))) solve_tile<T>(sh.colbuf, wb, lc, Ab, lda, i, P, lane);
    if (i == P) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        acc_t c = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c = MM<T>::mma(dW[l15 * LDC + 4 * q + lq], sh.rowbuf[(4 * q + lq) * LDR + 16 * j + l15], c);
        S[r][j] = c;
#pragma unroll
        for (int e = 0; e < 4; ++e) ur[MM<T>::row_of(lane, e) * LDR + 16 * j + l15] = c[e];
      }
      acc_t c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c[e] = dW[MM<T>::row_of(lane, e) * LDC + l15];
        ur[MM<T>::row_of(lane, e) * LDR + 16 * P + l15] = c[e];
      }
      S[r][P] = c;
    }
  }
  lds_barrier();
  // trailing update of the owned rows below the panel; column P+1 first (it is published for the next solve)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int i = tile_row(u, r);
    if (i > P) {
      T a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = -lc[(16 * i + l15) * LDC + 4 * q + lq];
      S[r][P] = acc_t{0, 0, 0, 0};
      if constexpr (P + 1 < 8) {
        if (i > P + 1 || !(P + 1 < np)) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            S[r][P + 1] = MM<T>::mma(a[q], lc[(16 * (P + 1) + l15) * LDC + 4 * q + lq], S[r][P + 1]);
        }
        if (i > P + 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sh.colbuf[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = S[r][P + 1][e];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j > i || j == P + 1) continue;
        if (j > P) {
#pragma unroll
          for (int q = 0; q < 4; ++q) S[r][j] = MM<T>::mma(a[q], lc[(16 * j + l15) * LDC + 4 * q + lq], S[r][j]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) S[r][j] = MM<T>::mma(a[q], ur[(4 * q + lq) * LDR + 16 * j + l15], S[r][j]);
        }
      }
      if (i == P + 1) {  // V_(P+1)j, j <= P, is final: image for the solve of panel P+1
#pragma unroll
        for (int j = 0; j <= P; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) sh.rowbuf[MM<T>::row_of(lane, e) * LDR + 16 * j + l15] = S[r][j][e];
      }
    }
  }
}

template <typename T, int P>
__device__ __forceinline__ void diag_panel(Shared<T>& sh, T* __restrict__ Ab, int lda, int lane, double& lg,
                                           int& bad) {
  const int l15 = lane & 15, lq = lane >> 4;
  T d[16], wv[16];
  if constexpr (P == 0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = Ab[(size_t)l15 * lda + c];
  } else {
    wait_ge<false>(&sh.flagA, P + 1, bad);
    const T* dA = sh.diagA[P & 1];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = dA[l15 * LDC + c];
  }
  int badk = -1;
  diag16<T>(d, wv, l15, badk);
  if (lq == 0) {
    T* dW = sh.diagW[P & 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) dW[i * LDC + l15] = wv[i];
  }
  post(&sh.flagW, P + 1);
  if (badk >= 0 && bad == 0) bad = 16 * P + badk + 1;
  if (lq == 0) {
    T dkk = d[0];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c <= l15) Ab[(size_t)(16 * P + l15) * lda + 16 * P + c] = d[c];
      if (c == l15) dkk = d[c];
    }
    lg += log((double)dkk);
  }
  lds_barrier();
}
}  // namespace leaf5

template <typename T>
__global__ __launch_bounds__(256, 1) void leaf5_kernel(T* __restrict__ A, long long sA, int lda, T* __restrict__ W,
                                                       long long sW, int ldw, int off, double* __restrict__ logdet,
                                                       int* __restrict__ info, int nvalid) {
  using acc_t = typename MM<T>::acc_t;
  using namespace leaf5;
  __shared__ Shared<T> sh;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15;
  T* Ab = A + (size_t)blockIdx.x * sA;
  T* Wb = W + (size_t)blockIdx.x * sW;
  // panels that start at or beyond nvalid are identity padding: L = I, W = I, already in place
  const int np = max(1, min(8, (nvalid + 15) >> 4));
  if (t == 0) {
    sh.flagA = 0;
    sh.flagW = 0;
  }
  __syncthreads();
  int bad = 0;
  if (w == 0) {
    double lg = 0.0;
    diag_panel<T, 0>(sh, Ab, lda, lane, lg, bad);
    if (np > 1) diag_panel<T, 1>(sh, Ab, lda, lane, lg, bad);
    if (np > 2) diag_panel<T, 2>(sh, Ab, lda, lane, lg, bad);
    if (np > 3) diag_panel<T, 3>(sh, Ab, lda, lane, lg, bad);
    if (np > 4) diag_panel<T, 4>(sh, Ab, lda, lane, lg, bad);
    if (np > 5) diag_panel<T, 5>(sh, Ab, lda, lane, lg, bad);
    if (np > 6) diag_panel<T, 6>(sh, Ab, lda, lane, lg, bad);
    if (np > 7) diag_panel<T, 7>(sh, Ab, lda, lane, lg, bad);
    lg = wave_sum(lg);
    if (lane == 0) {
      if (bad) atomicCAS(info + blockIdx.x, 0, off + bad);
      atomicAdd(logdet + blockIdx.x, lg);
    }
    return;
  }
  const int u = w - 1;
  acc_t S[3][8];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int i = tile_row(u, r);
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
    if (i > 0) {  // image of A_i0 for the solve of panel 0
#pragma unroll
      for (int e = 0; e < 4; ++e) sh.colbuf[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = S[r][0][e];
    }
  }
  update_panel<T, 0>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 1) update_panel<T, 1>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 2) update_panel<T, 2>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 3) update_panel<T, 3>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 4) update_panel<T, 4>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 5) update_panel<T, 5>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 6) update_panel<T, 6>(S, sh, Ab, lda, u, lane, np, bad);
  if (np > 7) update_panel<T, 7>(S, sh, Ab, lda, u, lane, np, bad);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int i = tile_row(u, r);
    if (i < 0) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * i + MM<T>::row_of(lane, e), col = 16 * j + l15;
        Wb[(size_t)row * ldw + col] = (j <= i) ? S[r][j][e] : (T)0;
      }
  }
  if (bad && lane == 0) atomicCAS(info + blockIdx.x, 0, off + 1);  // a hand-off timed out (never expected)
}

inline int g_leaf_version = 5;  // 5: pipelined leaf5 (default), 3: the barrier-per-phase leaf3

template <typename T>
inline void launch_leaf(hipStream_t st, int batch, T* A, long long sA, int lda, T* W, long long sW, int ldw, int off,
                        double* logdet, int* info, int nvalid) {
  if (g_leaf_version == 3)
    hipLaunchKernelGGL((leaf3_kernel<T>), dim3(batch), dim3(256), 0, st, A, sA, lda, W, sW, ldw, off, logdet, info,
                       nvalid);
  else
    hipLaunchKernelGGL((leaf5_kernel<T>), dim3(batch), dim3(256), 0, st, A, sA, lda, W, sW, ldw, off, logdet, info,
                       nvalid);
}

}  // namespace gpc

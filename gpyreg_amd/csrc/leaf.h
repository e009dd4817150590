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

// 1/sqrt(x) to working precision from the hardware estimate plus Newton steps in residual form
// (y += y (1/2 - (x/2) y^2): three dependent operations per step).  The pivot's square root and reciprocal sit on
// the serial critical path of every elimination step; diag16 issues these operations one at a time between the
// (independent) eliminations of the previous pivot.
__device__ __forceinline__ double rsq_estimate(double x) { return __builtin_amdgcn_rsq(x); }
__device__ __forceinline__ float rsq_estimate(float x) { return __builtin_amdgcn_rsqf(x); }
template <typename T>
struct RsqSteps {
  static constexpr int N = std::is_same<T, double>::value ? 6 : 3;  // two Newton steps for fp64, one for fp32
  T h, y, a, e;
  __device__ __forceinline__ void start(T x) {
    y = rsq_estimate(x);
    h = (T)0.5 * x;
  }
  __device__ __forceinline__ void step(int i) {  // i = 0 .. N-1, in order
    switch (i % 3) {
      case 0: a = h * y; break;
      case 1: e = __builtin_fma(-a, y, (T)0.5); break;
      default: y = __builtin_fma(y, e, y); break;
    }
  }
};
template <typename T>
__device__ __forceinline__ T fast_rsqrt(T x) {
  RsqSteps<T> n;
  n.start(x);
#pragma unroll
  for (int i = 0; i < RsqSteps<T>::N; ++i) n.step(i);
  return n.y;
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
// dj -= dk * bcast_J(dk);  wj -= bcast_J(dk) * wk.  NOP: dk was written by the instruction just before (the
// VALU-write -> DPP-read hazard needs two wait states); later eliminations of the same pivot read a dk that is long settled.
template <int J, bool NOP>
__device__ __forceinline__ void elim16(double& dj, double& wj, double dk, double wk) {
  double t;
  if constexpr (NOP)
    asm volatile(
        "s_nop 1\n\tv_mov_b64_dpp %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f64 %0, -%3, %2, %0\n\tv_fma_f64 %1, -%2, %4, %1"
        : "+v"(dj), "+v"(wj), "=&v"(t)
        : "v"(dk), "v"(wk), "n"(J));
  else
    asm volatile(
        "v_mov_b64_dpp %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f64 %0, -%3, %2, %0\n\tv_fma_f64 %1, -%2, %4, %1"
        : "+v"(dj), "+v"(wj), "=&v"(t)
        : "v"(dk), "v"(wk), "n"(J));
}
template <int J, bool NOP>
__device__ __forceinline__ void elim16(float& dj, float& wj, float dk, float wk) {
  float t;
  if constexpr (NOP)
    asm volatile(
        "s_nop 1\n\tv_mov_b32_dpp %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %0, -%3, %2, %0\n\tv_fma_f32 %1, -%2, %4, %1"
        : "+v"(dj), "+v"(wj), "=&v"(t)
        : "v"(dk), "v"(wk), "n"(J));
  else
    asm volatile(
        "v_mov_b32_dpp %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
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
  //
  // Look-ahead on the pivot: row k+1 is eliminated first, its diagonal entry is then final, and the reciprocal
  // square root of pivot k+1 (one estimate + 3 or 6 dependent operations, ~110 cycles when issued back to back)
  // is issued one operation at a time BETWEEN the remaining eliminations of pivot k, which do not depend on it.
  // __builtin_amdgcn_sched_barrier pins the interleave (the elimination blocks are volatile asm and keep their
  // order by themselves; the Newton operations are ordinary code the scheduler would otherwise group).
#pragma unroll
  for (int i = 0; i < 16; ++i) wv[i] = (l15 == i) ? (T)1 : (T)0;
  T s = bcast16<0>(d[0]);
  if (!(s > (T)0)) badk = 0;
  T r = fast_rsqrt(s);
  static_for<0, 16>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    d[k] = d[k] * r;    // column k of L (lane k: L_kk = s / sqrt(s))
    wv[k] = wv[k] * r;  // W[k][c] = (delta_kc - sum_{m<k} L_km W[m][c]) / L_kk
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (k + 1 < 16) {
      elim16<k + 1, true>(d[k + 1], wv[k + 1], d[k], wv[k]);
      const T sn = bcast16<k + 1>(d[k + 1]);
      if (!(sn > (T)0) && badk < 0) badk = k + 1;
      RsqSteps<T> n;
      n.start(sn);
      __builtin_amdgcn_sched_barrier(0);
      static_for<k + 2, 16>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        elim16<j, false>(d[j], wv[j], d[k], wv[k]);
        if constexpr (j - (k + 2) < RsqSteps<T>::N) n.step(j - (k + 2));
        __builtin_amdgcn_sched_barrier(0);
      });
      constexpr int done = 16 - (k + 2) < 0 ? 0 : 16 - (k + 2);
      static_for<(done < RsqSteps<T>::N ? done : RsqSteps<T>::N), RsqSteps<T>::N>([&](auto ic) { n.step(decltype(ic)::value); });
      r = n.y;
    }
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

// STABLE (the mode of the jitter retries, plan.h): the panel solve as a product with the explicit inverse W_PP has an
// error of cond(L_PP) eps instead of eps -- on a numerically singular matrix that perturbs the trailing pivots by far
// more than their size, and the factorization fails where a triangular solve (LAPACK) succeeds
// (tools/jitter_model.py).  One step of refinement against the factor itself,
//   L_iP += (A_iP - L_iP L_PP^T) W_PP^T,
// restores the accuracy of a solve.  myL: per-wave image of L_PP, zeros above the diagonal.
template <typename T, int P, bool STABLE>
__device__ __forceinline__ void leaf3_panel(typename MM<T>::acc_t (&S)[2][8], T* __restrict__ colbuf,
                                            T* __restrict__ lcol, T* __restrict__ rowbuf,
                                            T* __restrict__ urow, T* __restrict__ myW, T* __restrict__ myL,
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
    if constexpr (STABLE) {
#pragma unroll
      for (int c = 0; c < 16; ++c) myL[l15 * LDC + c] = (c <= l15) ? d[c] : (T)0;
    }
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
      if constexpr (STABLE) {
        // rows 16 i .. 16 i + 15 of lcol / colbuf are this wave's alone in this phase: scratch for the relayouts
        // (accumulator layout -> A-operand fragments); LDS operations of one wave complete in order
#pragma unroll
        for (int e = 0; e < 4; ++e) lcol[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = c[e];
        __builtin_amdgcn_wave_barrier();
        acc_t res = S[r][P];  // A_iP
#pragma unroll
        for (int q = 0; q < 4; ++q)
          res = MM<T>::mma(-lcol[(16 * i + l15) * LDC + 4 * q + lq], myL[l15 * LDC + 4 * q + lq], res);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) colbuf[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = res[e];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q)
          c = MM<T>::mma(colbuf[(16 * i + l15) * LDC + 4 * q + lq], myW[l15 * LDC + 4 * q + lq], c);
        __builtin_amdgcn_wave_barrier();
      }
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
template <typename T, bool STABLE>
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
  __shared__ T dgL[STABLE ? 4 : 1][16 * LDC];
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
  leaf3_panel<T, 0, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 16) leaf3_panel<T, 1, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 32) leaf3_panel<T, 2, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 48) leaf3_panel<T, 3, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 64) leaf3_panel<T, 4, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 80) leaf3_panel<T, 5, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 96) leaf3_panel<T, 6, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);
  if (nvalid > 112) leaf3_panel<T, 7, STABLE>(S, colbuf, lcol, rowbuf, urow, dgW[w], dgL[STABLE ? w : 0], Ab, lda, w, lane, lg, bad);

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
  if constexpr (STABLE) {
    // L11 of an enclosing node becomes a GEMM operand (plan.h, stable mode): the part of this tile above the diagonal
    // still holds entries of the input matrix and must read as zero
    for (int idx = t; idx < TILE * TILE; idx += 256) {
      const int row = idx / TILE, col = idx % TILE;
      if (col > row) Ab[(size_t)row * lda + col] = (T)0;
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
// the 36 accumulator slots (tile rows {7,2,1}, {6,3,0}, {5,4}: compile-time roles, so every tile loop is
// straight-line code) and do all the MFMA work of panel P while the diag wave is already inside the chain of
// panel P+1: the owner of tile row P+1 first computes L_(P+1)P and the one tile the next chain needs
// (A_(P+1)(P+1) -= L L^T, from its own data), signals, and only then joins the rest of the solve and the trailing
// update.  Hand-offs are LDS flags (release / acquire at workgroup scope); "L_iP, W_Pj published" is separated from
// their readers by an LDS arrival counter of the three update waves -- the diag wave never waits for anything but
// its next tile, and no hardware barrier is used after the start.  Images read across that point are double
// buffered by panel parity.  Tile products of one phase are interleaved over their independent accumulators
// (k-step outermost); every accumulator still sees the MFMA sequence of leaf3 on the same operands, so the results
// are bit-identical (tests/test_gpu_kernels.py).
//
// tools/leaf_probe.hip defines GPC_LEAF_TRACE: lane 0 of every wave stamps the shader clock at the phase boundaries
#ifdef GPC_LEAF_TRACE
__device__ long long* g_leaf_trace = nullptr;  // [4 waves][8 panels][8 stamps]
#define LEAF_TS(W, P, k) \
  do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_leaf_trace[((W) * 8 + (P)) * 8 + (k)] = clock64(); } while (0)
#else
#define LEAF_TS(W, P, k) do { } while (0)
#endif

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
  T diagL[2][16 * LDC];    // L_PP from the diag wave (its owner among the update waves writes it to memory)
  T piv[TILE];             // diag L, for the log-determinant
  double logv[8][16];
  int flagA;               // = P + 1 once A_PP is in diagA[P & 1]
  int flagW;               // = P + 1 once W_PP is in diagW[P & 1]
  int arrivedL;            // update waves that have published their L_iP of panel P: 3 (P + 1) when all have
  int arrivedW;            // = P + 1 once the owner of tile row P has published W_Pj, j <= P
  int rowReady;            // = 1 once V_7j, j < 7, is in rowbuf (the last panel's products are dealt to all three waves)
};
// tile rows of update wave U (0..2), slot r (0..2); -1: none
constexpr int trow(int U, int r) { return r == 0 ? 7 - U : (r == 1 ? 2 + U : 1 - U); }
constexpr int slot_of(int U, int i) { return trow(U, 0) == i ? 0 : (trow(U, 1) == i ? 1 : (trow(U, 2) == i ? 2 : -1)); }

// Every wave reaches the end of every wait: a hand-off that never comes (a bug, not a data condition) ends the wait
// after ~0.1 s instead of hanging the queue, and sets `tmo`.  The kernel then reports LEAF_TIMEOUT in `info` -- a code
// of its own, which the host turns into an ERROR (gpcore.hip: Pipe::run), never into a jitter retry: the results of
// such a leaf are garbage, not "not positive definite".  After its first time-out a wave no longer waits at all.
template <bool SLEEP>
__device__ __forceinline__ void wait_ge(int* f, int v, int& tmo) {
  if (tmo) return;
  int n = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
    if (SLEEP) __builtin_amdgcn_s_sleep(1);
    if (++n > (1 << 21)) {
      tmo = 1;
      break;
    }
  }
}
__device__ __forceinline__ void post(int* f, int v) {
  __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <typename T, int U, int P>
__device__ __forceinline__ void update_panel(typename MM<T>::acc_t (&S)[3][8], Shared<T>& sh, T* __restrict__ Ab,
                                             int lda, T* __restrict__ Wb, int ldw, int lane, int& bad) {
  // (`bad` of an update wave: one of its waits timed out -- the update waves see no pivots)
  using acc_t = typename MM<T>::acc_t;
  const int l15 = lane & 15, lq = lane >> 4;
  T* lc = sh.lcol[P & 1];
  T* ur = sh.urow[P & 1];
  const T* dW = sh.diagW[P & 1];
  constexpr int RF = P + 1 < 8 ? slot_of(U, P + 1) : -1;  // slot of tile row P+1 (fast path) or -1
  constexpr int RP = slot_of(U, P);                        // slot of tile row P (final row of the inverse) or -1
  LEAF_TS(U + 1, P, 0);
  wait_ge<(RF < 0 && P < 7)>(&sh.flagW, P + 1, bad);  // the waves on the critical path poll without sleeping
  LEAF_TS(U + 1, P, 1);
  T wb[4];  // fragments of W_PP: B operand of A_iP W_PP^T and A operand of W_PP V_Pj
#pragma unroll
  for (int q = 0; q < 4; ++q) wb[q] = dW[l15 * LDC + 4 * q + lq];
  // fast path: the owner of tile row P+1 hands the next diagonal tile to the diag wave before anything else
  if constexpr (RF >= 0) {
    constexpr int i = P + 1;
    acc_t c = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) c = MM<T>::mma(sh.colbuf[(16 * i + l15) * LDC + 4 * q + lq], wb[q], c);
#pragma unroll
    for (int e = 0; e < 4; ++e) lc[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = c[e];
    __builtin_amdgcn_wave_barrier();
    T b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = lc[(16 * i + l15) * LDC + 4 * q + lq];
#pragma unroll
    for (int q = 0; q < 4; ++q) S[RF][i] = MM<T>::mma(-b[q], b[q], S[RF][i]);
    T* dA = sh.diagA[i & 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) dA[MM<T>::row_of(lane, e) * LDC + l15] = S[RF][i][e];
    post(&sh.flagA, i + 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) Ab[(size_t)(16 * i + MM<T>::row_of(lane, e)) * lda + 16 * P + l15] = c[e];
  }
  LEAF_TS(U + 1, P, 2);
  // the other owned tile rows below the panel (k-step outermost over the independent tiles)
  {
    acc_t c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) c[r] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      static_for<0, 3>([&](auto rc) {
        constexpr int r = decltype(rc)::value, i = trow(U, r);
        if constexpr (i > P && r != RF) c[r] = MM<T>::mma(sh.colbuf[(16 * i + l15) * LDC + 4 * q + lq], wb[q], c[r]);
      });
    static_for<0, 3>([&](auto rc) {
      constexpr int r = decltype(rc)::value, i = trow(U, r);
      if constexpr (i > P && r != RF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * i + MM<T>::row_of(lane, e);
          lc[row * LDC + l15] = c[r][e];
          Ab[(size_t)row * lda + 16 * P + l15] = c[r][e];
        }
      }
    });
  }
  if (lane == 0) __hip_atomic_fetch_add(&sh.arrivedL, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  // the owner of tile row P: W_Pj = W_PP V_Pj (j < P) and W_PP itself -- the final row P of the inverse; the other
  // waves meanwhile update their A tiles, which need the L images only
  // Last panel: nothing is left to update, the chain has ended, and the seven products W_7j are all that stands
  // between it and the end of the kernel -- the two waves without rows take four of them and store their tiles
  // of the inverse themselves (LAST_SHARE columns stay with the owner).
  constexpr int LAST_SHARE = 3;
  if constexpr (P == 7 && RP < 0) {
    wait_ge<false>(&sh.rowReady, 1, bad);
    constexpr int J0 = LAST_SHARE + 2 * (U - 1);
    acc_t c[2];
    c[0] = c[1] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j) c[j] = MM<T>::mma(wb[q], sh.rowbuf[(4 * q + lq) * LDR + 16 * (J0 + j) + l15], c[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) Wb[(size_t)(16 * 7 + MM<T>::row_of(lane, e)) * ldw + 16 * (J0 + j) + l15] = c[j][e];
  }
  if constexpr (RP >= 0) {
    constexpr int NJ = P == 7 ? LAST_SHARE : P;  // tile columns j < NJ of row P are this wave's
    if constexpr (P > 0) {
      acc_t c[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) c[j] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NJ; ++j) c[j] = MM<T>::mma(wb[q], sh.rowbuf[(4 * q + lq) * LDR + 16 * j + l15], c[j]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        S[RP][j] = c[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) ur[MM<T>::row_of(lane, e) * LDR + 16 * j + l15] = c[j][e];
      }
    }
    acc_t c;
    const T* dL = sh.diagL[P & 1];
    T lpp[4];  // L_PP is read BEFORE this panel's last hand-off: the diag wave may reuse the buffer two chains later
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // all reads first: an LDS write between them makes every read a round trip
      const int row = MM<T>::row_of(lane, e);
      c[e] = dW[row * LDC + l15];
      lpp[e] = dL[row * LDC + l15];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) ur[MM<T>::row_of(lane, e) * LDR + 16 * P + l15] = c[e];
    S[RP][P] = c;
    post(&sh.arrivedW, P + 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = MM<T>::row_of(lane, e);
      if (l15 <= row) Ab[(size_t)(16 * P + row) * lda + 16 * P + l15] = lpp[e];
    }
    // row P of the inverse is final: to memory now, while the other waves update, instead of at the end of the kernel
    // (the last panel's tiles beyond LAST_SHARE are stored by the waves that computed them)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (P == 7 && j >= NJ && j < 7) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        Wb[(size_t)(16 * P + MM<T>::row_of(lane, e)) * ldw + 16 * j + l15] = (j <= P) ? S[RP][j][e] : (T)0;
    }
  }
  LEAF_TS(U + 1, P, 3);
  if constexpr (trow(U, 0) > P) {
    // trailing update of the owned rows below the panel, k-step outermost: the B fragment of tile column j is read
    // once per k-step and serves every owned row.  A tiles (j > P) first: they wait for the L images only.
    wait_ge<true>(&sh.arrivedL, 3 * (P + 1), bad);
    LEAF_TS(U + 1, P, 4);
    T a[3][4];
    static_for<0, 3>([&](auto rc) {
      constexpr int r = decltype(rc)::value, i = trow(U, r);
      if constexpr (i > P) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[r][q] = -lc[(16 * i + l15) * LDC + 4 * q + lq];
        S[r][P] = acc_t{0, 0, 0, 0};
      }
    });
#pragma unroll
    for (int q = 0; q < 4; ++q)
      static_for<P + 1, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j <= trow(U, 0)) {  // row slot 0 is the lowest owned row: the widest
          const T b = lc[(16 * j + l15) * LDC + 4 * q + lq];
          static_for<0, 3>([&](auto rc) {
            constexpr int r = decltype(rc)::value, i = trow(U, r);
            if constexpr (i > P && j <= i && !(r == RF && j == P + 1)) S[r][j] = MM<T>::mma(a[r][q], b, S[r][j]);
          });
        }
      });
    if constexpr (P + 1 < 8) {  // images of A_i(P+1) for the solve of the next panel
      static_for<0, 3>([&](auto rc) {
        constexpr int r = decltype(rc)::value, i = trow(U, r);
        if constexpr (i > P + 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sh.colbuf[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = S[r][P + 1][e];
        }
      });
    }
    // V tiles (j <= P) need the final row P of the inverse
    wait_ge<true>(&sh.arrivedW, P + 1, bad);
    LEAF_TS(U + 1, P, 5);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      static_for<0, P + 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const T b = ur[(4 * q + lq) * LDR + 16 * j + l15];
        static_for<0, 3>([&](auto rc) {
          constexpr int r = decltype(rc)::value, i = trow(U, r);
          if constexpr (i > P) S[r][j] = MM<T>::mma(a[r][q], b, S[r][j]);
        });
      });
    if constexpr (RF >= 0) {  // V_(P+1)j is final: image for the solve of the next panel
#pragma unroll
      for (int j = 0; j <= P; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) sh.rowbuf[MM<T>::row_of(lane, e) * LDR + 16 * j + l15] = S[RF][j][e];
      if constexpr (P == 6) post(&sh.rowReady, 1);
    }
  }
}

template <typename T, int U>
__device__ __forceinline__ void update_wave(Shared<T>& sh, T* __restrict__ Ab, int lda, T* __restrict__ Wb, int ldw,
                                            int lane, int np, int& bad) {
  using acc_t = typename MM<T>::acc_t;
  const int l15 = lane & 15;
  acc_t S[3][8];
  // column 0 and tile (1,1) first: the solve of panel 0 and the first fast path wait for nothing else
  static_for<0, 8>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    static_for<0, 3>([&](auto rc) {
      constexpr int r = decltype(rc)::value, i = trow(U, r);
      if constexpr (i >= 0) {
        if constexpr (j <= i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 16 * i + MM<T>::row_of(lane, e), col = 16 * j + l15;
            const T v = Ab[(size_t)row * lda + col];
            S[r][j][e] = (j < i || col <= row) ? v : (T)0;
          }
        } else {
          S[r][j] = acc_t{0, 0, 0, 0};
        }
      } else {
        S[r][j] = acc_t{0, 0, 0, 0};
      }
    });
  });
  static_for<0, 3>([&](auto rc) {  // image of A_i0 for the solve of panel 0
    constexpr int r = decltype(rc)::value, i = trow(U, r);
    if constexpr (i > 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sh.colbuf[(16 * i + MM<T>::row_of(lane, e)) * LDC + l15] = S[r][0][e];
    }
  });
  update_panel<T, U, 0>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 1) update_panel<T, U, 1>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 2) update_panel<T, U, 2>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 3) update_panel<T, U, 3>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 4) update_panel<T, U, 4>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 5) update_panel<T, U, 5>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 6) update_panel<T, U, 6>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  if (np > 7) update_panel<T, U, 7>(S, sh, Ab, lda, Wb, ldw, lane, bad);
  static_for<0, 3>([&](auto rc) {
    constexpr int r = decltype(rc)::value, i = trow(U, r);
    if constexpr (i >= 0) {
      if (i < np) return;  // rows that were a panel row went to memory when they became final (update_panel)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * i + MM<T>::row_of(lane, e), col = 16 * j + l15;
          Wb[(size_t)row * ldw + col] = (j <= i) ? S[r][j][e] : (T)0;
        }
      }
    }
  });
}

template <typename T, int P>
__device__ __forceinline__ void diag_panel(Shared<T>& sh, T* __restrict__ Ab, int lda, int lane, double& lg,
                                           int& bad, int& tmo) {
  const int l15 = lane & 15, lq = lane >> 4;
  T d[16], wv[16];
  LEAF_TS(0, P, 0);
  if constexpr (P == 0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = Ab[(size_t)l15 * lda + c];
  } else {
    wait_ge<false>(&sh.flagA, P + 1, tmo);
    const T* dA = sh.diagA[P & 1];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = dA[l15 * LDC + c];
  }
  int badk = -1;
  LEAF_TS(0, P, 1);
  diag16<T>(d, wv, l15, badk);
  LEAF_TS(0, P, 2);
  if (lq == 0) {
    T* dW = sh.diagW[P & 1];
    T* dL = sh.diagL[P & 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dW[i * LDC + l15] = wv[i];
      dL[l15 * LDC + i] = d[i];
    }
  }
  post(&sh.flagW, P + 1);
  LEAF_TS(0, P, 3);
  // off the critical path (the wave now waits for the next tile)
  if (badk >= 0 && bad == 0) bad = 16 * P + badk + 1;
  if (lq == 0) {
    T dkk = d[0];
#pragma unroll
    for (int c = 1; c < 16; ++c)
      if (c == l15) dkk = d[c];
    sh.piv[16 * P + l15] = dkk;
  }
  LEAF_TS(0, P, 4);
}
}  // namespace leaf5

// The whole pipelined leaf as a device function (all 256 threads of the block call it; the diag wave and the update
// waves return at different times -- the caller synchronises the block before it reads L / W): leaf5_kernel below, and
// the fused leaf + solves kernel of problems that ARE one leaf (leaf_solve_kernel).
template <typename T>
__device__ __forceinline__ void leaf5_body(leaf5::Shared<T>& sh, T* __restrict__ Ab, int lda, T* __restrict__ Wb, int ldw,
                                           int off, double* __restrict__ logdet_b, int* __restrict__ info_b, int nvalid,
                                           int fault) {
  using namespace leaf5;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // panels that start at or beyond nvalid are identity padding: L = I, W = I, already in place
  const int np = max(1, min(8, (nvalid + 15) >> 4));
  if (threadIdx.x == 0) {
    sh.flagA = 0;
    sh.flagW = 0;
    sh.arrivedL = 0;
    sh.arrivedW = 0;
    sh.rowReady = 0;
  }
  __syncthreads();
  int bad = 0;
  if (w == 0) {
    double lg = 0.0;
    int tmo = 0;
    diag_panel<T, 0>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 1) diag_panel<T, 1>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 2) diag_panel<T, 2>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 3) diag_panel<T, 3>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 4) diag_panel<T, 4>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 5) diag_panel<T, 5>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 6) diag_panel<T, 6>(sh, Ab, lda, lane, lg, bad, tmo);
    if (np > 7) diag_panel<T, 7>(sh, Ab, lda, lane, lg, bad, tmo);
    // sum(log diag L): two logarithms per lane instead of eight in a row, summed per column in panel order
    // (the order of leaf3) by the first row of 16 lanes
    __builtin_amdgcn_wave_barrier();
    {
      const int l15 = lane & 15, lq = lane >> 4;
      if (lq < np) sh.logv[lq][l15] = log((double)sh.piv[16 * lq + l15]);
      if (lq + 4 < np) sh.logv[lq + 4][l15] = log((double)sh.piv[16 * (lq + 4) + l15]);
      __builtin_amdgcn_wave_barrier();
      if (lq == 0)
        for (int P = 0; P < np; ++P) lg += sh.logv[P][l15];
    }
    lg = wave_sum(lg);
    if (lane == 0) {
      if (tmo) atomicOr(info_b, LEAF_TIMEOUT);
      if (bad) atomicCAS(info_b, 0, off + bad);
      atomicAdd(logdet_b, lg);
    }
    return;
  }
  // test hook (gpc_set_option "leaf_fault"): one update wave never shows up, so every hand-off it owes times out
  if (fault && w == 1) return;
  if (w == 1)
    update_wave<T, 0>(sh, Ab, lda, Wb, ldw, lane, np, bad);
  else if (w == 2)
    update_wave<T, 1>(sh, Ab, lda, Wb, ldw, lane, np, bad);
  else
    update_wave<T, 2>(sh, Ab, lda, Wb, ldw, lane, np, bad);
  if (bad && lane == 0) atomicOr(info_b, LEAF_TIMEOUT);  // a hand-off timed out (never expected)
}

template <typename T>
__global__ __launch_bounds__(256, 1) void leaf5_kernel(T* __restrict__ A, long long sA, int lda, T* __restrict__ W,
                                                       long long sW, int ldw, int off, double* __restrict__ logdet,
                                                       int* __restrict__ info, int nvalid, int fault) {
  __shared__ leaf5::Shared<T> sh;
  leaf5_body<T>(sh, A + (size_t)blockIdx.x * sA, lda, W + (size_t)blockIdx.x * sW, ldw, off, logdet + blockIdx.x,
                info + blockIdx.x, nvalid, fault);
}

// ---- problems that are ONE leaf (N <= 128: the single evaluations of the slice sampler and of the optimiser on small
// training sets, fit's 1024-point design; f_min_fill.py:174-176, slice_sample.py:442): the factorization above and the
// two triangular products that follow it in ONE launch.  The pipeline used to be leaf | z = W r | z.z (| W^T z partial |
// its sum): three to four dependent launches of a few microseconds of work each behind a 20 us leaf.  Here the block
// that factored the matrix reads its own W back (L2 / L1 hot) and finishes the job:
//     z = W r,   quad = z.z,   alpha = W^T z / sl  (gradient evaluations only)
// and, for an evaluation without gradient, writes [logdet | quad | info] of its sample straight into the host's pinned
// landing block (`land`, mapped memory), so no download kernel follows -- and, when `flag` is given, announces the call's
// completion there as well (round 6: the host polls that word instead of waiting for the stream).   grid = (batch), 256 threads.
template <typename T>
__global__ __launch_bounds__(256, 1) void leaf_solve_kernel(T* __restrict__ A, long long sA, int lda, T* __restrict__ W,
                                                            long long sW, int ldw, double* __restrict__ logdet,
                                                            int* __restrict__ info, int nvalid, int fault,
                                                            const double* __restrict__ r_all, double* __restrict__ z_all,
                                                            double* __restrict__ quad_all, double* __restrict__ alpha_all,
                                                            const double* __restrict__ sp_all, int sp_stride, int sp_sl,
                                                            double* __restrict__ land, int cnt,
                                                            unsigned long long* __restrict__ flag = nullptr,
                                                            unsigned long long seq = 0, int* __restrict__ done_ctr = nullptr) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  __shared__ leaf5::Shared<T> sh;
  __shared__ double zs[TILE], red[2][TILE], sh4[4];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  T* Wb = W + (size_t)b * sW;
  // r first (it may live in mapped host memory: the round trip hides under the factorization)
  const double* r = r_all + (size_t)b * TILE;
  double rk[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) rk[e] = lane * VEC + e < TILE ? r[lane * VEC + e] : 0.0;
  leaf5_body<T>(sh, A + (size_t)b * sA, lda, Wb, ldw, 0, logdet + b, info + b, nvalid, fault);
  __syncthreads();  // L and W of this block are in memory (and visible to the block: __syncthreads orders global stores)
  // z_i = sum_k W[i][k] r[k]: one wave per row, all 32 rows of a wave in flight at once (one L2 round trip; W is zero
  // above the diagonal: whole rows).  The 32 lane sums are folded across the wave by halving: after a step with
  // offset o a lane keeps half of its values, so 32 values cost 16 + 8 + 4 + 2 + 1 exchanges and one last one.
  {
    double s[32];
    const int i0 = w * 32;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      s[u] = 0.0;
      if (lane * VEC < TILE) {
        const vec_t wv = *reinterpret_cast<const vec_t*>(Wb + (size_t)(i0 + u) * ldw + lane * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[u] = fma((double)wv[e], rk[e], s[u]);
      }
    }
    // halving steps: offsets 32, 16, 8, 4, 2 -- lane bit set: keep the upper half of the values, send the lower
#pragma unroll
    for (int n = 32, o = 32; n > 1; n >>= 1, o >>= 1) {
      const bool up = (lane & o) != 0;
#pragma unroll
      for (int u = 0; u < n / 2; ++u) {
        const double keep = up ? s[u + n / 2] : s[u];
        const double give = up ? s[u] : s[u + n / 2];
        s[u] = keep + __shfl_xor(give, o, 64);
      }
    }
    // lane l now holds the sum over the lanes that agree with it in bit 0 of value index bits(l >> 1): one more step
    const double v = s[0] + __shfl_xor(s[0], 1, 64);
    // value index: bit 4 from lane bit 5, ..., bit 0 from lane bit 1
    const int idx = ((lane >> 5) & 1) << 4 | ((lane >> 4) & 1) << 3 | ((lane >> 3) & 1) << 2 | ((lane >> 2) & 1) << 1 |
                    ((lane >> 1) & 1);
    if ((lane & 1) == 0) zs[i0 + idx] = v;
  }
  __syncthreads();
  double q = t < TILE ? zs[t] * zs[t] : 0.0;
  q = block_sum_256(q, sh4);
  if (t < TILE) z_all[(size_t)b * TILE + t] = zs[t];
  if (alpha_all) {
    // alpha_k = sum_{i >= k} W[i][k] z_i / sl: a thread per column and half of the rows (coalesced rows of W)
    const int k = t & (TILE - 1), h = t >> 7;
    double s = 0.0;
    for (int i = 64 * h; i < 64 * h + 64; ++i) s = fma((double)Wb[(size_t)i * ldw + k], zs[i], s);
    red[h][k] = s;
    __syncthreads();
    if (t < TILE) alpha_all[(size_t)b * TILE + t] = (red[0][t] + red[1][t]) / sp_all[(size_t)b * sp_stride + sp_sl];
  }
  if (t == 0) {
    quad_all[b] = q;
    if (land) {  // [logdet[cnt] | quad[cnt] | info[cnt] (ints)]: this sample's three values, straight to the host
      const double ld = __hip_atomic_load(logdet + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int inf = __hip_atomic_load(info + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      land[b] = ld;
      land[cnt + b] = q;
      reinterpret_cast<int*>(land + 2 * (size_t)cnt)[b] = inf;
      if (flag) {
        // The host does not wait for the end-of-kernel signal of the runtime (completion packet, cache write-back,
        // wake-up: several microseconds after the last instruction) but watches `flag`, a word of the same coherent host
        // block: every block makes its three values visible system-wide, counts itself, and the block that finds itself
        // the last one publishes this call's sequence number.
        __threadfence_system();
        const int before = __hip_atomic_fetch_add(done_ctr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == cnt - 1) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

inline int g_leaf_version = 5;  // 5: pipelined leaf5 (default), 3: the barrier-per-phase leaf3
inline int g_leaf_fault = 0;    // test hook: leaf5 runs with a missing update wave (its hand-offs time out)

template <typename T>
inline void launch_leaf(hipStream_t st, int batch, T* A, long long sA, int lda, T* W, long long sW, int ldw, int off,
                        double* logdet, int* info, int nvalid, bool stable = false) {
  if (stable)
    hipLaunchKernelGGL((leaf3_kernel<T, true>), dim3(batch), dim3(256), 0, st, A, sA, lda, W, sW, ldw, off, logdet, info,
                       nvalid);
  else if (g_leaf_version == 3)
    hipLaunchKernelGGL((leaf3_kernel<T, false>), dim3(batch), dim3(256), 0, st, A, sA, lda, W, sW, ldw, off, logdet, info,
                       nvalid);
  else
    hipLaunchKernelGGL((leaf5_kernel<T>), dim3(batch), dim3(256), 0, st, A, sA, lda, W, sW, ldw, off, logdet, info,
                       nvalid, g_leaf_fault);
}

}  // namespace gpc

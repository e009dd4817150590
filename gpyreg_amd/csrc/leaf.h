// leaf.h -- 128 x 128 Cholesky factor AND its inverse in one workgroup (batched).
//
// This is the "diagonal potrf + trsm seed" of the blocked factorization: for the
// diagonal block D it produces L (D = L L^T, written to the lower triangle of the
// block in A) and W = L^-1 (full 128 x 128 tile written to the W buffer, zeros above
// the diagonal), plus sum(log diag L) and the LAPACK-style info flag.
//
// Algorithm: a right-looking elimination in which the 128 x 128 block lives in
// REGISTERS (each of the 256 threads owns an 8 x 8 set of entries, cyclically
// distributed: rows ty+16a, cols tx+16b) and only the pivot column and pivot row
// travel through LDS (2 x 1 KB per step, double buffered, one barrier per step).
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

template <typename T, int JB>
__device__ __forceinline__ void leaf_steps(T (&M)[8][8], T* __restrict__ colbuf,
                                           T* __restrict__ rowbuf, T* __restrict__ dbuf,
                                           T* __restrict__ Aout, int lda, int tx, int ty, int& bad) {
  for (int jj = 0; jj < 16; ++jj) {
    const int j = JB * 16 + jj;
    T* cb = colbuf + (j & 1) * TILE;
    T* rb = rowbuf + (j & 1) * TILE;
    if (tx == jj) {
#pragma unroll
      for (int a = 0; a < 8; ++a) cb[ty + 16 * a] = M[a][JB];
    }
    if (ty == jj) {
#pragma unroll
      for (int b = 0; b < 8; ++b) rb[tx + 16 * b] = M[JB][b];
    }
    __syncthreads();
    const T piv = cb[j];
    if (!(piv > (T)0) && bad == 0) bad = j + 1;
    const T rinv = fast_rsqrt(piv);
    const T d = piv * rinv;
    if (tx == jj && ty == jj) dbuf[j] = d;

    T c[8], v[8];
#pragma unroll
    for (int a = JB; a < 8; ++a) c[a] = cb[ty + 16 * a] * rinv;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int k = tx + 16 * b;
      if (b > JB)
        v[b] = cb[k] * rinv;
      else if (b < JB)
        v[b] = rb[k] * rinv;
      else
        v[b] = (tx > jj) ? cb[k] * rinv : ((tx == jj) ? rinv : rb[k] * rinv);
    }
    // pivot-column owners: emit L[:, j] and recycle the register for W[:, j]
    if (tx == jj) {
#pragma unroll
      for (int a = JB; a < 8; ++a) {
        const int i = ty + 16 * a;
        if (i > j) {
          Aout[(size_t)i * lda + j] = c[a];
          M[a][JB] = (T)0;
        } else if (i == j) {
          Aout[(size_t)j * lda + j] = d;
        }
      }
    }
    // rank-1 update of every row below the pivot
#pragma unroll
    for (int a = JB; a < 8; ++a) {
      const bool active = (a > JB) || (ty > jj);
      if (active) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          // columns right of the pivot block still hold A: only its lower triangle is
          // ever read, so blocks strictly above the diagonal are skipped
          if (b > JB && b > a) continue;
          M[a][b] -= c[a] * v[b];
        }
      }
    }
    // pivot row becomes the final row j of W (zeros right of the diagonal)
    if (ty == jj) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int k = tx + 16 * b;
        M[JB][b] = (k <= j) ? v[b] : (T)0;
      }
    }
  }
}

// grid = (batch); A, W point at the top-left of the diagonal block of sample 0.
template <typename T>
__global__ __launch_bounds__(256) void leaf_kernel(T* __restrict__ A, long long sA, int lda,
                                                   T* __restrict__ W, long long sW, int ldw,
                                                   int off, double* __restrict__ logdet,
                                                   int* __restrict__ info) {
  __shared__ T colbuf[2 * TILE];
  __shared__ T rowbuf[2 * TILE];
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

  int bad = 0;
  leaf_steps<T, 0>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 1>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 2>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 3>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 4>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 5>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 6>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);
  leaf_steps<T, 7>(M, colbuf, rowbuf, dbuf, Ab, lda, tx, ty, bad);

#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = ty + 16 * a, k = tx + 16 * b;
      Wb[(size_t)i * ldw + k] = M[a][b];
    }
  __syncthreads();
  const double lg = block_sum_256(t < TILE ? log((double)dbuf[t]) : 0.0, red4);
  if (t == 0) {
    if (bad) atomicCAS(info + blockIdx.x, 0, off + bad);
    atomicAdd(logdet + blockIdx.x, lg);
  }
}

}  // namespace gpc

// plan.h -- host-side launch plan of the blocked factorization (batched over samples).
//
//   potrf_inv : A = L L^T  and  W = L^-1   (recursive; every product is gemm.h, the
//               128 x 128 diagonal blocks are leaf.h)
//   lauum     : Ainv = W^T W               (one triangular-aware GEMM launch)
//
// The recursion (verified on the CPU by tests/blocked_model.py, which executes this
// same plan with NumPy tiles) on a diagonal block split [n1 | n2]:
//   1. potrf_inv(A11)                      -> L11, W11
//   2. T21 = A21 * W11^T                   (the trsm, as a product; k <= column tile)
//   3. A22 -= T21 * T21^T                  (syrk, lower tiles)
//   4. potrf_inv(A22)                      -> L22, W22
//   5. U = T21 * W11 (into A21);  W21 = -W22 * U      (only if this block's inverse is needed)
//   6. A21 = T21                           (only where L21 must survive)
// Buffers: A (in: SPD lower; out: L), W (out), T (scratch; receives Ainv).  No launch
// aliases its output with an input.  Strictly-upper tiles of A, W, T are never read.
#pragma once
#include "blas1.h"
#include "gemm.h"
#include "leaf.h"

#include <algorithm>

namespace gpc {

#ifdef GPC_EXPERIMENTS
// Recording mode (dag.h): instead of launching, the plan appends every product and every leaf to a list, in launch
// order -- the tile-level dataflow schedule is derived from exactly the launches the stream-ordered schedule would
// have issued, so both run the same tile arithmetic.
struct PlanRecorder {
  struct Op {
    int kind;  // 0: product (g, akm, bkm), 1: leaf at diagonal offset `off`
    GemmArgs g;
    bool akm, bkm;
    int off;
  };
  std::vector<Op> ops;
  bool unsupported = false;  // the plan asked for something the dataflow kernel does not do (L21 copies, stable leaves)
};
#endif

template <typename T>
struct Factor {
#ifdef GPC_EXPERIMENTS
  PlanRecorder* rec = nullptr;  // experiments build: record the plan instead of issuing it (dag.h)
#else
  static constexpr void* rec = nullptr;  // (the product library issues every launch; `if (rec)` folds away)
#endif
  hipStream_t st;
  int batch;
  int npad;
  int nvalid = 1 << 30;  // rows/columns below this index are real data, the rest identity padding
  T* A;
  T* W;
  T* Tm;
  long long sA, sW, sT;  // batch strides (elements)
  double* logdet;        // [batch], must be zeroed by the caller
  int* info;             // [batch], must be zeroed by the caller
  // zeroed device counters for persistent GEMM launches (gemm.h); one per launch, in order
  int* ctr = nullptr;
  int ctr_cap = 0, ctr_used = 0;
  // Deferred inverse products (GPC_DEFER_MIN > 0): U = T21 * W11 of a node of size >= defer_min does not depend
  // on the node's right child, whose factorization starts with latency-bound work (leaves, deep levels).  U is
  // launched on `side` as a persistent GEMM that keeps `reserve` CUs per XCD empty (gemm.h), the right child
  // proceeds on `st` (its small kernels land on the empty CUs), and W21 = -W22 * U joins the two.
  hipStream_t side = nullptr;
  hipEvent_t* evs = nullptr;  // pool of events for the fork / join pairs
  int nev = 0, ev_used = 0;
  int defer_min = 0;
  const unsigned short* reserve = nullptr;  // table of the CUs a deferred launch stays off (gemm.h: cu_reserve_bail)
  // debug (gpc_set_option "check_queues"): the persistent launches issued, for a check after the pipeline that
  // every tile queue was drained
  struct QueueCheck {
    int* slot;
    int ntiles, batch;
  };
  std::vector<QueueCheck>* qlog = nullptr;
#ifdef GPC_EXPERIMENTS
  bool reserve_all = false;  // see gemm(): all chip-filling launches of this pipeline are CU-reserving (independent pipelines)
  int reserve_min = 64;
  int reserved_small_bt = 128;  // tile of a reserved launch below the 128-tile threshold (64: independent pipelines)
#else
  static constexpr bool reserve_all = false;
  static constexpr int reserve_min = 64, reserved_small_bt = 128;
#endif
  bool dual_launch = g_dual_launch;  // syrk + inverse product of a node in one launch where both are small
  // Stable mode (the jitter retries, gpcore.hip: retry_failed): the trsm-as-a-product T21 = A21 W11^T has an error of
  // cond(L11) eps instead of eps, which on a numerically singular matrix makes the factorization fail where a
  // triangular solve (LAPACK, the reference) succeeds -- the device needed 1-2 decades more jitter than the reference
  // on 7 of 8 singular fixture samples and never less (tools/jitter_model.py reproduces it on the CPU).  One step of
  // refinement against the factor itself, T21 += (A21 - T21 L11^T) W11^T, at every node and in the leaf's panel
  // solve (leaf.h: STABLE) restores the accuracy of a solve; L21 is kept in A so that L11 is a complete operand.
  bool stable = false;
  // rows >= tail_row0 of A are still being built on the side stream (covfun.h: build_persist_kernel); the first
  // launch that touches them waits for ev_tail
  hipEvent_t ev_tail = nullptr;
  int tail_row0 = 1 << 30;
  void need_rows(int hi) {
    if (ev_tail && hi > tail_row0) {
      chk(hipStreamWaitEvent(st, ev_tail, 0));
      ev_tail = nullptr;
    }
  }
  double flops = 0;      // algorithmic flops issued (tile-exact)
  int launches = 0;
  hipError_t err = hipSuccess;

  T* blk(T* base, int r, int c) const { return base + (size_t)r * npad + c; }

  GemmArgs make_args(T* C, long long sC, const T* Aop, long long sAop, const T* Bop, long long sBop, int M, int N,
                     int K, double alpha, int beta, int klo, int khi, int lower) {
    GemmArgs g;
    g.A = Aop;
    g.B = Bop;
    g.C = C;
    g.sA = sAop;
    g.sB = sBop;
    g.sC = sC;
    g.lda = g.ldb = g.ldc = npad;
    g.M = M;
    g.N = N;
    g.K = K;
    g.alpha = alpha;
    g.beta = beta;
    g.klo = klo;
    g.khi = khi;
    g.lower_only = lower;
    g.tiles_n = N / TILE;
    flops += gemm_flops(g, batch);
    return g;
  }

  void gemm(T* C, long long sC, const T* Aop, long long sAop, const T* Bop, long long sBop, int M,
            int N, int K, bool akm, bool bkm, double alpha, int beta, int klo, int khi, int lower,
            hipStream_t on = nullptr, const unsigned short* rsv = nullptr) {
    GemmArgs g;
    g.A = Aop;
    g.B = Bop;
    g.C = C;
    g.sA = sAop;
    g.sB = sBop;
    g.sC = sC;
    g.lda = g.ldb = g.ldc = npad;
    g.M = M;
    g.N = N;
    g.K = K;
    g.alpha = alpha;
    g.beta = beta;
    g.klo = klo;
    g.khi = khi;
    g.lower_only = lower;
    g.tiles_n = N / TILE;
    flops += gemm_flops(g, batch);
    ++launches;
#ifdef GPC_EXPERIMENTS
    if (rec) {
      rec->ops.push_back({0, g, akm, bkm, 0});
      return;
    }
#endif
    int* slot = nullptr;  // CTR_STRIDE counters per persistent launch
    // only launches that will run in the persistent form take a slot: a factorization of npad = 8192 or
    // 16384 has 250-500 launches, and the big ones (which need the slots) come last in the order
    const long long tm = M / TILE, tn = N / TILE;
    const long long blocks128 = (lower ? tm * (tm + 1) / 2 : tm * tn) * batch;
    // independent pipelines (gpcore.hip: Pipe::run, option "indep"): every launch of at least reserve_min 128-tiles stays off
    // the reserved CUs, on THIS stream -- the other pipelines' leaves and small launches run there meanwhile
    if (!rsv && reserve_all && reserve && blocks128 >= reserve_min) rsv = reserve;
    const bool persistent = rsv || (blocks128 >= g_small_launch_blocks && blocks128 > g_block_slots - g_persist_spare);
    if (persistent && ctr && ctr_used + CTR_STRIDE <= ctr_cap) {
      slot = ctr + ctr_used;
      ctr_used += CTR_STRIDE;
      if (qlog) qlog->push_back({slot, (int)(lower ? tm * (tm + 1) / 2 : tm * tn), batch});
    }
    hipError_t e = launch_gemm<T>(on ? on : st, g, akm, bkm, batch, 0, slot, rsv, reserved_small_bt);
    if (e != hipSuccess && err == hipSuccess) err = e;
  }

  void chk(hipError_t e) {
    if (e != hipSuccess && err == hipSuccess) err = e;
  }

  void potrf_inv(int off, int n, bool need_inv, bool keep_L) {
    keep_L = keep_L || stable;
    if (n == TILE) {
#ifdef GPC_EXPERIMENTS
      if (rec) {
        if (stable) rec->unsupported = true;
        rec->ops.push_back({1, GemmArgs{}, false, false, off});
        flops += (2.0 / 3.0) * TILE * (double)TILE * TILE * batch;
        ++launches;
        return;
      }
#endif
      need_rows(off + TILE);
      launch_leaf<T>(st, batch, blk(A, off, off), sA, npad, blk(W, off, off), sW, npad, off, logdet, info,
                     std::max(0, std::min(TILE, nvalid - off)), stable);
      flops += (2.0 / 3.0) * TILE * (double)TILE * TILE * batch;
      ++launches;
      return;
    }
    const int q = n / TILE;
    const int n1 = (q / 2) * TILE, n2 = n - n1;
    const int o1 = off, o2 = off + n1;
    potrf_inv(o1, n1, true, keep_L);
    need_rows(o2 + n2);
    // 2. T21 = A21 * W11^T
    gemm(blk(Tm, o2, o1), sT, blk(A, o2, o1), sA, blk(W, o1, o1), sW, n2, n1, n1, false, false, 1.0,
         0, KLO_ZERO, KHI_COL, 0);
    if (stable) {
      // 2b. R = A21 - T21 * L11^T (in place: A21 is only ever the C operand), T21 += R * W11^T
      gemm(blk(A, o2, o1), sA, blk(Tm, o2, o1), sT, blk(A, o1, o1), sA, n2, n1, n1, false, false, -1.0, 1, KLO_ZERO,
           KHI_COL, 0);
      gemm(blk(Tm, o2, o1), sT, blk(A, o2, o1), sA, blk(W, o1, o1), sW, n2, n1, n1, false, false, 1.0, 1, KLO_ZERO,
           KHI_COL, 0);
    }
    const bool defer = need_inv && side && defer_min > 0 && n >= defer_min && ev_used + 2 <= nev;
    // 3. A22 -= T21 * T21^T  -- and, where both are small launches, 5a. U = T21 * W11 -> A21 in the same launch:
    // it waits for T21 only (A21 has been consumed by step 2), and a launch less per node is ~5 us less
    bool u_done = false;
    {
      GemmArgs gs = make_args(blk(A, o2, o2), sA, blk(Tm, o2, o1), sT, blk(Tm, o2, o1), sT, n2, n2, n1, -1.0, 1,
                              KLO_ZERO, KHI_FULL, 1);
      const long long syrk_tiles = (long long)(n2 / TILE) * (n2 / TILE + 1) / 2 * batch;
      if (need_inv && !defer && dual_launch && !rec && gemm_is_small(gs, batch) && !(reserve_all && syrk_tiles >= reserve_min)) {
        GemmArgs gu = make_args(blk(A, o2, o1), sA, blk(Tm, o2, o1), sT, blk(W, o1, o1), sW, n2, n1, n1, 1.0, 0,
                                KLO_COL, KHI_FULL, 0);
        if (gemm_is_small(gu, batch)) {
          hipError_t e = launch_gemm_dual_small<T>(st, gs, gu, batch);
          if (e != hipSuccess && err == hipSuccess) err = e;
          ++launches;
          u_done = true;
        } else {
          flops -= gemm_flops(gu, batch);
        }
      }
      if (!u_done) {
        flops -= gemm_flops(gs, batch);
        gemm(blk(A, o2, o2), sA, blk(Tm, o2, o1), sT, blk(Tm, o2, o1), sT, n2, n2, n1, false, false, -1.0, 1,
             KLO_ZERO, KHI_FULL, 1);
      }
    }
    hipEvent_t ev_join = nullptr;
    if (defer) {
      // 5a early, on the side stream: U = T21 * W11 -> A21 (reads T21, W11: untouched by the right child)
      hipEvent_t ev_fork = evs[ev_used++];
      ev_join = evs[ev_used++];
      chk(hipEventRecord(ev_fork, st));
      chk(hipStreamWaitEvent(side, ev_fork, 0));
      gemm(blk(A, o2, o1), sA, blk(Tm, o2, o1), sT, blk(W, o1, o1), sW, n2, n1, n1, false, true, 1.0, 0,
           KLO_COL, KHI_FULL, 0, side, reserve);
      chk(hipEventRecord(ev_join, side));
    }
    potrf_inv(o2, n2, need_inv, keep_L);
    if (need_inv) {
      // 5a. U = T21 * W11  -> A21
      if (defer)
        chk(hipStreamWaitEvent(st, ev_join, 0));
      else if (!u_done)
        gemm(blk(A, o2, o1), sA, blk(Tm, o2, o1), sT, blk(W, o1, o1), sW, n2, n1, n1, false, true, 1.0, 0,
             KLO_COL, KHI_FULL, 0);
      // 5b. W21 = -W22 * U
      gemm(blk(W, o2, o1), sW, blk(W, o2, o2), sW, blk(A, o2, o1), sA, n2, n1, n2, false, true, -1.0,
           0, KLO_ZERO, KHI_ROW, 0);
    }
    // L21 back into A: only where L must survive as the Cholesky factor (posteriors).  The blocked forward solve
    // of an NLL-only evaluation reads L21 where it was computed, in the scratch (forward_solve below).
#ifdef GPC_EXPERIMENTS
    if (keep_L && rec) rec->unsupported = true;
#endif
    if (keep_L && !rec) {
      dim3 grid((n1 + 64 * MM<T>::VEC - 1) / (64 * MM<T>::VEC), n2 / 32, batch), block(64, 4);
      hipLaunchKernelGGL((rect_copy_kernel<T>), grid, block, 0, st, (const T*)blk(Tm, o2, o1), sT, npad,
                         blk(A, o2, o1), sA, npad, n2, n1);
      ++launches;
    }
  }

  // ---- NLL only: A = L L^T at N^3/3 ---------------------------------------------------------------------------
  // An evaluation without gradient (the design stage of fit, f_min_fill.py:174-176, and the slice sampler,
  // slice_sample.py:442) needs log det and L^-1 r, not the inverse.  potrf_inv(.., need_inv = false) still inverts
  // every LEFT child completely (its T21 = A21 W11^T wants W11): 0.381 N^3 flop.  Here only diagonal blocks of at
  // most `nll_block` rows get their inverse (2 N nll_block^2 / 3 flop in all); above that size the panel
  // T21 = A21 L11^-T is a BLOCKED triangular solve against the factor itself,
  //     X_a = A21_a L_a^-T ;  A21_c -= X_a L_ca^T ;  X_c = A21_c L_c^-T        (L11 = [[L_a, 0], [L_ca, L_c]]),
  // recursively, the base case a product with the inverse of a block of nll_block rows.  Same flops as the product
  // with the full inverse, but no U / W21 products above nll_block rows.  L_ca is read where the inner node left it,
  // in the scratch (its own T21).
  int nll_block = 0;  // 0: off (potrf_inv with need_inv = false)

  void trsm_nll(int r0, int m, int c0, int n) {
    if (n <= nll_block) {
      gemm(blk(Tm, r0, c0), sT, blk(A, r0, c0), sA, blk(W, c0, c0), sW, m, n, n, false, false, 1.0, 0, KLO_ZERO,
           KHI_COL, 0);
      if (stable) {  // refinement against the factor of the block, as in potrf_inv (L of the block is complete in A)
        gemm(blk(A, r0, c0), sA, blk(Tm, r0, c0), sT, blk(A, c0, c0), sA, m, n, n, false, false, -1.0, 1, KLO_ZERO,
             KHI_COL, 0);
        gemm(blk(Tm, r0, c0), sT, blk(A, r0, c0), sA, blk(W, c0, c0), sW, m, n, n, false, false, 1.0, 1, KLO_ZERO,
             KHI_COL, 0);
      }
      return;
    }
    const int q = n / TILE;
    const int n1 = (q / 2) * TILE, n2 = n - n1;
    trsm_nll(r0, m, c0, n1);
    gemm(blk(A, r0, c0 + n1), sA, blk(Tm, r0, c0), sT, blk(Tm, c0 + n1, c0), sT, m, n2, n1, false, false, -1.0, 1,
         KLO_ZERO, KHI_FULL, 0);
    trsm_nll(r0, m, c0 + n1, n2);
  }

  void potrf_nll(int off, int n) {
    if (n <= nll_block || n == TILE) {
      potrf_inv(off, n, true, false);
      return;
    }
    const int q = n / TILE;
    const int n1 = (q / 2) * TILE, n2 = n - n1;
    const int o1 = off, o2 = off + n1;
    potrf_nll(o1, n1);
    need_rows(o2 + n2);
    trsm_nll(o2, n2, o1, n1);
    gemm(blk(A, o2, o2), sA, blk(Tm, o2, o1), sT, blk(Tm, o2, o1), sT, n2, n2, n1, false, false, -1.0, 1, KLO_ZERO,
         KHI_FULL, 1);
    potrf_nll(o2, n2);
  }

  // z = L^-1 r after potrf_nll: blocks of at most nll_block rows multiply by their inverse, above that the solve
  // splits like the factorization and eliminates with L21, in the scratch
  void forward_solve_nll(int off, int n, double* r, double* z) {
    if (n <= nll_block || n == TILE) {
      hipLaunchKernelGGL((trmv_kernel<T>), dim3(n / 4, batch), dim3(256), 0, st, (const T*)W, sW, npad,
                         (const double*)r, npad, z, off);
      ++launches;
      return;
    }
    const int q = n / TILE;
    const int n1 = (q / 2) * TILE, n2 = n - n1;
    forward_solve_nll(off, n1, r, z);
    hipLaunchKernelGGL((gemv_sub_kernel<T>), dim3(n2 / 4, batch), dim3(256), 0, st, (const T*)Tm, sT, npad,
                       (const double*)z, r, npad, off + n1, off, n1);
    ++launches;
    forward_solve_nll(off + n1, n2, r, z);
  }

#ifdef GPC_EXPERIMENTS
  // ---- right-looking panels with one panel of look-ahead (round 3) ------------------------------------------------
  // The recursion above exposes every latency-bound stretch of a child (its leaves, its deep-level products) on the
  // critical path.  Here the matrix is factored in panels of `rl_panel` rows, right-looking:
  //     D_k  potrf_inv of the diagonal block (L_kk, W_kk: the recursion above, at most rl_panel rows)
  //     P_k  panel  L[below, k] = A[below, k] W_kk^T                                    -> scratch
  //     N_k  trailing update of the NEXT block column only   A[below, k+1] -= L[below, k] L[k+1, k]^T
  //     R_k  trailing update of the rest (lower tiles)       A[i, j] -= L[i, k] L[j, k]^T,  j >= k+2
  // and R_k runs on the side stream as a CU-reserving persistent launch while D_{k+1} (latency-bound, on the reserved
  // CUs) and P_{k+1} (which reads block column k+1 only) already proceed on the main stream; N_{k+1} joins.  The
  // chip-filling work (sum of the R_k: almost all of the N^3/3) hides the diagonal blocks.  L below the diagonal
  // blocks lives in the scratch, as with the recursion.
  int rl_panel = 0;
  bool rl_lookahead = true;  // false: R_k on the main stream, whole chip (same arithmetic, launches in order)

  void potrf_rl() {
    const int P = rl_panel;
    hipEvent_t pending = nullptr;  // join event of the R launch still in flight on the side stream
    for (int o = 0; o < npad; o += P) {
      const int nb = std::min(P, npad - o), rest = npad - o - nb;
      need_rows(o + nb);
      potrf_inv(o, nb, true, false);
      if (rest == 0) break;
      need_rows(npad);
      gemm(blk(Tm, o + nb, o), sT, blk(A, o + nb, o), sA, blk(W, o, o), sW, rest, nb, nb, false, false, 1.0, 0,
           KLO_ZERO, KHI_COL, 0);
      const int nb2 = std::min(P, rest), rest2 = rest - nb2;
      if (pending) {  // R_{k-1} writes the block columns N_k and R_k are about to update
        chk(hipStreamWaitEvent(st, pending, 0));
        pending = nullptr;
      }
      // N_k as a full rectangle: the tiles above the diagonal of its first block are dead writes (never read)
      gemm(blk(A, o + nb, o + nb), sA, blk(Tm, o + nb, o), sT, blk(Tm, o + nb, o), sT, rest, nb2, nb, false, false,
           -1.0, 1, KLO_ZERO, KHI_FULL, 0);
      if (rest2 > 0) {
        const int o2 = o + nb + nb2;
        const bool ahead = rl_lookahead && side && reserve && ev_used + 2 <= nev;
        if (ahead) {
          hipEvent_t ev_fork = evs[ev_used++];
          pending = evs[ev_used++];
          chk(hipEventRecord(ev_fork, st));
          chk(hipStreamWaitEvent(side, ev_fork, 0));
        }
        gemm(blk(A, o2, o2), sA, blk(Tm, o2, o), sT, blk(Tm, o2, o), sT, rest2, rest2, nb, false, false, -1.0, 1,
             KLO_ZERO, KHI_FULL, 1, ahead ? side : nullptr, ahead ? reserve : nullptr);
        if (ahead) chk(hipEventRecord(pending, side));
      }
    }
    if (pending) chk(hipStreamWaitEvent(st, pending, 0));
  }

  // z = L^-1 r after potrf_rl: per panel, multiply by the diagonal block's inverse and eliminate with the panel below
  void forward_solve_rl(double* r, double* z) {
    const int P = rl_panel;
    for (int o = 0; o < npad; o += P) {
      const int nb = std::min(P, npad - o), rest = npad - o - nb;
      hipLaunchKernelGGL((trmv_kernel<T>), dim3(nb / 4, batch), dim3(256), 0, st, (const T*)W, sW, npad,
                         (const double*)r, npad, z, o);
      ++launches;
      if (rest > 0) {
        hipLaunchKernelGGL((gemv_sub_kernel<T>), dim3(rest / 4, batch), dim3(256), 0, st, (const T*)Tm, sT, npad,
                           (const double*)z, r, npad, o + nb, o, nb);
        ++launches;
      }
    }
  }

#endif  // GPC_EXPERIMENTS

  // z = L^-1 r after potrf_inv(.., need_inv): blocks that own their full inverse multiply
  // by W, the others split like the factorization and eliminate with L21 (kept in A for
  // exactly those blocks).  r is consumed (updated in place); r, z: [batch][npad] doubles.
  bool low_regs = false;  // forward_solve of the whole matrix with the kernel that fits beside the W^T W launch
  void forward_solve(int off, int n, bool need_inv, double* r, double* z) {
    if (need_inv || n == TILE) {
      if (low_regs && off == 0 && n == npad)
        hipLaunchKernelGGL((trmv_low_kernel<T>), dim3(n / 16, batch), dim3(256), 0, st, (const T*)W, sW, npad,
                           (const double*)r, npad, z);
      else
        hipLaunchKernelGGL((trmv_kernel<T>), dim3(n / 4, batch), dim3(256), 0, st, (const T*)W, sW, npad,
                           (const double*)r, npad, z, off);
      ++launches;
      return;
    }
    const int q = n / TILE;
    const int n1 = (q / 2) * TILE, n2 = n - n1;
    forward_solve(off, n1, true, r, z);
    // L21 of a block without its own inverse is still in the scratch (potrf_inv step 2; nothing has written that
    // block since), and in A as well when keep_L copied it there: same values
    hipLaunchKernelGGL((gemv_sub_kernel<T>), dim3(n2 / 4, batch), dim3(256), 0, st, (const T*)Tm, sT, npad,
                       (const double*)z, r, npad, off + n1, off, n1);
    ++launches;
    forward_solve(off + n1, n2, need_inv, r, z);
  }

  // Ainv (lower tiles, into `out`) = W^T W
  void lauum(T* out, long long sOut) {
    gemm(out, sOut, W, sW, W, sW, npad, npad, npad, true, true, 1.0, 0, KLO_ROW, KHI_FULL, 1);
  }
};

}  // namespace gpc

/*
 * gpcore.h -- C ABI of libgpcore.so, the MI355X (gfx950) dense Gaussian-process core.
 *
 * The reference (acerbilab/gpyreg) is pure Python and has no FFI; its boundary for
 * this path is the duck-typed plugin protocol GP.__init__(D, covariance, mean, noise)
 * (gaussian_process.py:43-49).  The entry points below are what a ctypes binding on
 * the reference side would call in place of the NumPy/SciPy bodies cited per function
 * (see INTEGRATION.md for the stub).  Plain pointers and sizes only; all arrays are
 * caller-owned host memory, row-major (C order) float64 unless stated; the library
 * copies inputs to the device and copies results back before returning.
 *
 * Return value: 0 = OK; -1 = HIP error, -2 = usage error, -3 = internal error (a wave hand-off inside a
 * 128 x 128 leaf factorization timed out; the results of that call are invalid) -- text via gpc_last_error;
 * per-sample numerical failure (a matrix that is not positive definite after the reference's ten jitter
 * levels, gaussian_process.py:2413-2421, :2450-2453) is reported in info[] (>0), never as a return code.
 *
 * Thread-safety: one gpc_ctx per host thread / per device; calls on one ctx serialize.
 */
#ifndef GPCORE_H
#define GPCORE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpc_ctx gpc_ctx;   /* one per (process, device): streams, workspace, X, y */
typedef struct gpc_post gpc_post; /* device-resident posteriors of one hyp batch        */

/* covariance families: covariance_functions.py:131 (SE), :189 (Matern), :288 (RQ-ARD),
 * isotropic_covariance_functions.py:164 (SE iso), :86 (Matern iso). */
enum { GPC_K_SE = 0, GPC_K_MATERN = 1, GPC_K_RQ = 2, GPC_K_SE_ISO = 3, GPC_K_MATERN_ISO = 4 };
/* arithmetic type of the factorization (kernel build and reductions are always f64) */
enum { GPC_F64 = 0, GPC_F32 = 1 };

/* ---- lifetime ---------------------------------------------------------------- */
int gpc_create(int device, gpc_ctx** out);
void gpc_destroy(gpc_ctx* ctx);
/* last error text of ctx (or of the failed gpc_create when ctx == NULL) */
const char* gpc_last_error(const gpc_ctx* ctx);
/* "gfx950 ..." style description of the device and library build */
const char* gpc_device_info(gpc_ctx* ctx);

/* ---- training data (GP.update stores X, y: gaussian_process.py:846-862) -------- */
/* X: N x D, y: N.  Kept resident in HBM until the next gpc_set_data. */
int gpc_set_data(gpc_ctx* ctx, const double* X, const double* y, int N, int D);

/* number of covariance hyperparameters (covariance_functions.py:59-73, :291-292;
 * isotropic_covariance_functions.py:14-28) */
int gpc_cov_count(int kernel_id, int D);
/* Largest N the dense stages accept for a dtype: a MEMORY-BUDGET answer (round 6) -- the largest multiple of 128 whose
 * three padded N x N slabs of one sample (matrix / factor, inverse factor, scratch) fit in 80 % of the current device's
 * memory (fp64 on a 288 GB MI355X: 101 504; fp32: 143 488).  The reference factorizes whatever fits host memory
 * (gaussian_process.py:2415-2417, :2477-2484); so does this library with device memory.  (Rounds 1-5 answered 16384 /
 * 23168: one 32-bit byte offset spanned a k-major operand panel.  The GEMM now advances a 64-bit base per k-slab.)
 * The batch entry points return -2 above it.  Without a visible device the answer assumes 288 GB.                       */
int gpc_max_n(int dtype);

/* ---- covariance.compute() (covariance_functions.py:135-186, :221-285, :301-367;
 *      isotropic_covariance_functions.py:104-161, :173-221) ------------------------
 * Xstar == NULL, diag == 0 : K is N x N; if dK != NULL it receives N x N x cov_N.
 * Xstar != NULL            : K is N x M (cross covariance), dK must be NULL.
 * diag != 0                : K is N (x1): the self-covariance diagonal.            */
int gpc_kernel(gpc_ctx* ctx, int kernel_id, int degree, const double* hyp_cov,
               const double* X, int N, int D, const double* Xstar, int M, int diag,
               double* K, double* dK);

/* ---- GP.__core_computation(hyp, 1, want_grad) for S hyperparameter vectors -------
 * (gaussian_process.py:2357-2512).  The covariance part runs on the device from
 * hyp_cov; mean and noise plugins are evaluated by the caller (they are O(N*D)
 * boundary plugins, mean_functions.py / noise_functions.py) and passed as arrays:
 *   hyp_cov  S x cov_N
 *   m        S x N           mean function values
 *   sn2      S x (sn2_is_vector ? N : 1)   noise variance (noise_functions.py:249-278)
 *   dm       S x N x mean_N  (want_grad && mean_N > 0, else NULL)
 *   dsn2     S x (sn2_is_vector ? N : 1) x noise_N (want_grad && noise_N > 0)
 * Outputs:
 *   nlz      S               negative log marginal likelihood
 *   dnlz     S x (cov_N+noise_N+mean_N), order [cov | noise | mean] (:2367-2369)
 *   sn2_mult S               final jitter multiplier (power of 10, :2413-2421)
 *   L_chol   S               1 if min(sn2) >= 1e-6 (:2404)
 *   info     S               0 ok; >0: still not positive definite after 10 tries
 *                            (the caller raises LinAlgError, :2450-2453)            */
int gpc_nll_batch(gpc_ctx* ctx, int kernel_id, int degree, int dtype, int S,
                  const double* hyp_cov, const double* m, const double* sn2,
                  int sn2_is_vector, int want_grad, const double* dm, int mean_N,
                  const double* dsn2, int noise_N, double* nlz, double* dnlz,
                  double* sn2_mult, int* L_chol, int* info);

/* ---- the same call for the stock ZeroMean / ConstantMean (mean_functions.py:82-131, :210-260): the mean is ONE value
 * per sample, m0[S] (mean_N = 1; NULL with mean_N = 0 for the zero mean), and its derivative is all ones -- so neither
 * m (S x N) nor dm (S x N x 1) crosses the bus: r = y - m0 is formed on the device from the resident y, and with a scalar
 * noise model so is the diagonal term.  Results are those of gpc_nll_batch with m[s][i] = m0[s], dm = 1, to the bit.     */
int gpc_nll_batch_cm(gpc_ctx* ctx, int kernel_id, int degree, int dtype, int S, const double* hyp_cov,
                     const double* m0, int mean_N, const double* sn2, int sn2_is_vector, int want_grad,
                     const double* dsn2, int noise_N, double* nlz, double* dnlz, double* sn2_mult,
                     int* L_chol, int* info);

/* ---- the same for ANY covariance object: caller-provided K and dK -------------------------------
 * The reference calls whatever object it was given -- `covariance.compute(hyp, X, compute_grad)`
 * (gaussian_process.py:2388-2390; AbstractKernel, covariance_functions.py:9-20).  A kernel this
 * library does not know is evaluated by the caller; the factorization, solves and the gradient
 * contraction still run on the device:
 *   K        S x N x N (row-major; symmetric)
 *   dk_plane callback: fill plane[N*N] (row-major) with dK[:, :, p] of sample `sample`; called once
 *            per (sample, p < cov_N) when want_grad, so the (N, N, cov_N) tensor is streamed one
 *            plane at a time and never resident on the device.  Return 0, or nonzero to abort.
 * Everything else (m, sn2, dm, dsn2, outputs, jitter escalation) as gpc_nll_batch; dnlz is ordered
 * [cov (cov_N) | noise | mean].                                                                     */
typedef int (*gpc_dk_plane_fn)(void* user, int sample, int p, double* plane);
int gpc_nll_batch_K(gpc_ctx* ctx, int dtype, int S, int cov_N, const double* K, gpc_dk_plane_fn dk_plane,
                    void* user, const double* m, const double* sn2, int sn2_is_vector, int want_grad,
                    const double* dm, int mean_N, const double* dsn2, int noise_N, double* nlz,
                    double* dnlz, double* sn2_mult, int* L_chol, int* info);
/* Posteriors from caller-provided K (GP.update with a user-defined kernel).  Use gpc_predict_K with
 * them; gpc_post_fetch / gpc_post_free as usual.                                                   */
int gpc_posterior_batch_K(gpc_ctx* ctx, int dtype, int S, const double* K, const double* m,
                          const double* sn2, int sn2_is_vector, gpc_post** post, double* sn2_mult,
                          int* L_chol, int* info);
/* Predictive products from caller-provided cross covariances Ks (S x N x M) -- works for posteriors
 * of either origin:
 *   fmu[j*S + s] = Ks_s[:, j] . alpha_s
 *   fq [j*S + s] = -colsum(V*V) (L_chol) or +colsum(Ks * (L Ks)): ADD kss to get s2 (:1752-1764); may be NULL
 *   cov[s]       = Kss_s - V^T V  or  Kss_s + Ks^T (L Ks)   (M x M; needs Kss, S x M x M); may be NULL  */
int gpc_predict_K(gpc_post* post, int M, const double* Ks, const double* Kss, double* fmu, double* fq,
                  double* cov);

/* ---- GP.__core_computation(hyp, 0, 0) -> Posterior, for S vectors
 *      (gaussian_process.py:2514-2521; GP.update loop :870-884) ---------------------
 * The factors stay in HBM inside *post (freed by gpc_post_free).                   */
int gpc_posterior_batch(gpc_ctx* ctx, int kernel_id, int degree, int dtype, int S,
                        const double* hyp_cov, const double* m, const double* sn2,
                        int sn2_is_vector, gpc_post** post, double* sn2_mult,
                        int* L_chol, int* info);
/* Posterior fields of sample s (gaussian_process.py:2568-2586).  Any pointer may be
 * NULL.  alpha: N.  sW: N.  L: N x N row-major holding, when L_chol, the LOWER
 * factor Lo with Lo Lo^T = (K + mult*Sigma)/sl -- the reference's upper factor is
 * its transpose (a free NumPy view) -- else -(K + mult*Sigma)^-1 (:2441-2448).      */
int gpc_post_fetch(gpc_post* post, int s, double* alpha, double* sW, double* L);
int gpc_post_free(gpc_post* post);

/* ---- GP.predict K* solves (gaussian_process.py:1741-1764) --------------------------
 * xstar: M x D.  For every posterior sample s:
 *   fmu[j*S + s]  = Ks^T alpha                (caller adds the mean function m*)
 *   fs2[j*S + s]  = kss - colsum(V*V)   or   kss + colsum(Ks * (L Ks))   (unclamped) */
int gpc_predict(gpc_post* post, const double* xstar, int M, double* fmu, double* fs2);

/* ---- rank-one append of ONE training point to resident posteriors (GP.update fast path,
 *      gaussian_process.py:750-844; scalar noise) ------------------------------------------------
 * Call gpc_set_data with the extended X (N+1 rows; the new point last) and y first.
 *   m_star[s]   mean function of sample s at the new point
 *   sn2_star[s] noise variance of sample s at the new point (noise.compute(hyp, x_new, y_new, 0))
 * High-noise samples (L_chol): l = W Ks, sqrt_arg = sn2_eff^2 + kss sn2_eff - l.l (:784-788); the
 * factor, its inverse and alpha get their new last row in O(N^2) (:800-817).  Low-noise samples
 * (Posterior.L = -inv): the rank-one update of -inv of :819-827.  The storage of EVERY sample grows
 * to N+1.  ok[s] = 1 if sample s was appended; ok[s] = 0 (sqrt_arg <= 0, a failed factorization,
 * or a noise value that is not the fitted scalar) leaves sample s stale: the caller recomputes
 * exactly those samples with gpc_post_recompute -- the reference's per-posterior fallback
 * (full_updates, :789-798 and :866-869).                                                        */
int gpc_post_append(gpc_post* post, const double* m_star, const double* sn2_star, double y_new, int* ok);
/* Recompute samples idx[0..cnt) of a resident posterior set in place from the context's current
 * data (__core_computation(hyp, 0, 0) for those samples, :866-869).  Arrays as gpc_posterior_batch,
 * one row per listed sample.                                                                      */
int gpc_post_recompute(gpc_post* post, int cnt, const int* idx, const double* hyp_cov, const double* m,
                       const double* sn2, int sn2_is_vector, double* sn2_mult, int* L_chol, int* info);
/* The same two steps for a posterior set built from caller-provided covariances (gpc_posterior_batch_K): the
 * reference's rank-one path calls self.covariance.compute whatever the object is (gaussian_process.py:771-772),
 * so the caller hands over what that call returns.  Ks: S x n (row s = k_s(X_old, x_new), the n = N - 1 points the
 * posterior was built on; gpc_set_data has been called with the extended X, y), kss: S (k_s(x_new, x_new)).
 * gpc_post_recompute_K: K = cnt x N x N matrices of the listed samples on the extended data.              */
int gpc_post_append_K(gpc_post* post, const double* Ks, const double* kss, const double* m_star,
                      const double* sn2_star, double y_new, int* ok);
int gpc_post_recompute_K(gpc_post* post, int cnt, const int* idx, const double* K, const double* m,
                         const double* sn2, int sn2_is_vector, double* sn2_mult, int* L_chol, int* info);

/* ---- GP.predict_full (gaussian_process.py:1603-1650) -------------------------------------
 * fmu[j*S + s] = Ks^T alpha;  cov[s] (M x M, row-major) = K** - V^T V  or  K** + Ks^T (L Ks)
 * (the caller symmetrises and adds noise, :1647-1659).                                    */
int gpc_predict_full(gpc_post* post, const double* xstar, int M, double* fmu, double* cov);

/* ---- GP.quad: Bayesian quadrature products (gaussian_process.py:1908-1966), SE kernels ----
 * mu, sigma: M x D (means and standard deviations of the Gaussian measures).  With z the
 * kernel mean vector of measure j under sample s:
 *   zalpha[j*S + s] = z . alpha                      (the caller adds the mean-function terms)
 *   zKz[j*S + s]    = z (K + sn2_eff I)^-1 z^T       (only if compute_var)                  */
int gpc_quad(gpc_post* post, const double* mu, const double* sigma, int M, int compute_var,
             double* zalpha, double* zKz);

/* ---- instrumentation -------------------------------------------------------------
 * GPU time (ms, hipEvent on the library's stream) of the last gpc_nll_batch /
 * gpc_posterior_batch: whole device section, and the part spent in the MFMA GEMM
 * launches + leaf factorizations (the N^3 work).  Calls below N_pad = 2048 record
 * these events only under gpc_set_option(ctx, "small_timing", 1): both are 0 otherwise. */
int gpc_last_timing(gpc_ctx* ctx, double* ms_total, double* ms_factor);
/* (after gpc_predict / gpc_predict_full / gpc_quad: ms_total = device time of the call, ms_factor = the
 * duration of its N^2 M product V = W Ks, the GEMM launch of gaussian_process.py:1752-1760)            */
/* The dominant single kernel of the last gpc_nll_batch with gradient: the W^T W ("lauum")
 * launch of gemm_kernel<T, true, true, ...>.  ms = its duration (hipEvents on the stream it
 * was launched on; the slowest sample group), flops = its algorithmic flops
 * (samples in that launch x N^3/3).                                                     */
int gpc_last_lauum_timing(gpc_ctx* ctx, double* ms, double* flops);
/* Tuning switches (also settable through the environment at gpc_create: GPC_GROUPS,
 * GPC_SMALL_BLOCKS, GPC_DEFER_MIN, GPC_DEFER_RESERVE): "groups" = sample groups on separate HIP
 * streams (1..8), "small_blocks" = launch size below which 64x64 tiles are used, "defer_min" = node
 * size from which the inverse product U = T21 W11 runs on a side stream (0 off, -1 auto),
 * "defer_reserve" = CUs per XCD that launch keeps empty (2 | 4 | 8 | 12).  Test hooks:
 * "start_mult_log10" = k starts the jitter escalation of every factorization at 10^k instead of 1
 * (gaussian_process.py:2402), "append_fail_mask" = bit s declares the rank-one append of sample s
 * unstable (:789-798).  "experiments" (get only): 1 when the loaded library is the experiments build.
 * Round 6, calls below N_pad = 2048 (single evaluations of the sampler and the optimiser on small training sets:
 * slice_sample.py:442, gaussian_process.py:1540): "small_poll" (default 1) = the call returns when a word that its last
 * launch writes into coherent host memory shows up, instead of waiting for the stream (bounded: after 0.15 - 2 ms it waits
 * for the stream after all); "small_timing" (default 0) = such calls record their timing events, so that gpc_last_timing
 * reports their device section (it reports 0 for them otherwise; from N_pad = 2048 on the events are always recorded).
 * "small_polled" / "small_synced" (get only): how many calls ended either way.                                         */
int gpc_set_option(gpc_ctx* ctx, const char* name, int value);
/* Current value of a tuning switch (so that a caller that changes one for a measurement can put it back). */
int gpc_get_option(gpc_ctx* ctx, const char* name, int* value);
/* fp64/fp32 MFMA issue-rate microbenchmark: achieved TFLOP/s of a register-resident
 * v_mfma_{f64,f32}_16x16x4 loop on all CUs (2 waves per SIMD), the shader cycles one
 * SIMD spends per MFMA, and the clock (GHz) the chip held while running it.  Used to
 * calibrate the roofline against what the silicon sustains rather than the datasheet. */
int gpc_mfma_peak(gpc_ctx* ctx, int dtype, double* tflops, double* cycles_per_mfma,
                  double* clock_ghz);

/* ---- test hooks (exercise one kernel through the ABI; used by tests/ only) --------
 * C[M x N] = beta*C + alpha*op(A)op(B) with the library's tiled MFMA GEMM.
 * a_kmajor: A stored K x M (else M x K); b_kmajor: B stored K x N (else N x K).
 * M, N, K multiples of 128.  klo/khi/lower_only: per-tile k-range modes (see
 * gpyreg_amd/csrc/gemm.h); lower_only bit 0 = lower tiles only, 0x100 / 0x200 force the 64- / 128-tile kernel variant
 * (0x400, experiments build only: the 128 x 64 tile).                              */
int gpc_debug_gemm(gpc_ctx* ctx, int dtype, int M, int N, int K, int a_kmajor,
                   int b_kmajor, double alpha, int beta, int klo, int khi,
                   int lower_only, const double* A, const double* B, double* C);
/* In-LDS leaf: A (128 x 128 SPD, lower used) -> L (lower) and W = L^-1; logdet, info */
int gpc_debug_leaf(gpc_ctx* ctx, int dtype, const double* A, double* L, double* W,
                   double* logdet, int* info);
/* Blocked factorization of an n x n SPD matrix (any n): L, W = L^-1, Ainv (lower).   */
int gpc_debug_factor(gpc_ctx* ctx, int dtype, int n, const double* A, double* L,
                     double* W, double* Ainv, double* logdet, int* info);

/* Debug: wrapping-sum hash of every 128 x 128 tile of one workspace matrix as the LAST call left it (which: 0 = A, 1 = W,
 * 2 = T; sample: position in the last chunk); out[(npad/128)^2].  Finds the tile where two schedules differ.          */
int gpc_debug_workspace_hash(gpc_ctx* ctx, int dtype, int which, int sample, unsigned long long* out);
/* ---- experiments build only (hipcc -DGPC_EXPERIMENTS -> lib/libgpcore_exp.so; NOT part of the product library) --------
 * Schedules that were built, measured and rejected (tile-level dataflow graph, independent pipelines,
 * rectangular / eight-wave tiles, right-looking panels) and their gpc_set_option names live there;
 * gpc_get_option(ctx, "experiments") says which build is loaded.                                                        */
#ifdef GPC_EXPERIMENTS
/* The tile-task graph of the dataflow schedule (gpyreg_amd/csrc/dag.h) for an npad x npad factorization --
 * HOST ONLY, no device needed: tests/test_dag_model.py executes it with NumPy tiles in random valid orders.
 * plan: 0 = NLL only (blocked solves above nll_blk rows when nll_blk > 0), 1 = factor + inverse + W^T W,
 * 2 = factor + inverse.  counts[4] = tasks, edges, launches, leaves.  With tasks_out == NULL only the counts are
 * written.  tasks_out: 24 ints per task [is_leaf, tile, a_kmajor, b_kmajor, beta, C/A/B region as (buffer, r0, r1,
 * c0, c1) each, predecessors, first successor, successors, ring]; alpha_out: one double per task; succ_out: the
 * successor lists.  Returns 0, -1 (plan not supported), -2 (bad arguments), -3 (buffers too small).               */
int gpc_debug_dag(int npad, int plan, int nll_blk, int small_tiles, int* counts, int* tasks_out,
                  double* alpha_out, int* succ_out, int cap_tasks, int cap_edges);

#endif /* GPC_EXPERIMENTS */

#ifdef __cplusplus
}
#endif
#endif /* GPCORE_H */

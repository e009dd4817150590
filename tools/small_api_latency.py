"""wall clock per call of the posterior / predict entry points at the sizes gpyreg's own users work at
(N = 100 .. 1000, a handful of hyperparameter samples, a few test points)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

for N, S, M in [(200, 1, 1), (200, 8, 1), (200, 8, 100), (500, 8, 1), (500, 8, 100), (1000, 8, 100)]:
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    for _ in range(3):
        gp.update(hyp=hyp)
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.update(hyp=hyp)
    t_up = (time.perf_counter() - t0) / reps
    xs = np.random.default_rng(1).uniform(-3, 3, (M, X.shape[1]))
    for _ in range(3):
        gp.predict(xs)
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.predict(xs)
    t_pr = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.predict(xs, separate_samples=True)
    t_ps = (time.perf_counter() - t0) / reps
    print(f"N={N:5d} S={S} M={M:4d}: update {t_up*1e3:7.3f} ms   predict {t_pr*1e3:7.3f} ms   predict(separate) {t_ps*1e3:7.3f} ms", flush=True)

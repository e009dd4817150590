"""GP.update with ONE new point (the rank-one path of a Bayesian-optimisation loop) against a full recompute,
wall clock per call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

for N, S in ((200, 8), (500, 8), (1000, 8), (2000, 8)):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N + 40)
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X[:N], y_new=y[:N], hyp=hyp)
    ts = []
    for i in range(N, N + 30):
        t0 = time.perf_counter()
        gp.update(X_new=X[i:i + 1], y_new=y[i:i + 1])
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    gp.update(hyp=hyp)
    t_full = time.perf_counter() - t0
    mu, s2 = gp.predict(X[:5])
    print(f"N={N:5d} S={S}: one-point update {np.median(ts)*1e3:.3f} ms (max {max(ts)*1e3:.3f}); full recompute {t_full*1e3:.3f} ms", flush=True)

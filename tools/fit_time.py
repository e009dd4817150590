"""wall clock of GP.fit (no profiler) at a few sizes; slice_speculate 1 (sequential) and 4 (default)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpyreg_amd as gpr

for N, D in ((100, 2), (300, 3), (1000, 5), (2000, 5)):
    np.random.seed(3)
    X = np.random.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * np.random.normal(size=(N, 1))

    def make():
        gp = gpr.GP(D=D, covariance=gpr.covariance_functions.Matern(5), mean=gpr.mean_functions.ConstantMean(),
                    noise=gpr.noise_functions.GaussianNoise(constant_add=True))
        gp.set_priors({
            "covariance_log_outputscale": ("student_t", (0, np.log(10), 3)),
            "covariance_log_lengthscale": ("gaussian", (np.log(np.std(X, ddof=1)), np.log(10))),
            "noise_log_scale": ("gaussian", (np.log(1e-2), 1.0)),
            "mean_const": ("smoothbox", (float(np.min(y)), float(np.max(y)), 1.0)),
        })
        return gp

    make().fit(X=X, y=y, options={"n_samples": 10})
    res = {}
    for spec in (1, 4, 0):
        gp = make()
        np.random.seed(4)
        t0 = time.perf_counter()
        opts = {"n_samples": 10}
        if spec:
            opts["slice_speculate"] = spec
        hyp, _, _ = gp.fit(X=X, y=y, options=opts)
        res[spec] = (time.perf_counter() - t0, hyp)
    same = np.array_equal(res[1][1], res[4][1]) and np.array_equal(res[1][1], res[0][1])
    print(f"fit N={N:5d} D={D}: sequential {res[1][0]:.3f} s, speculate 4: {res[4][0]:.3f} s, default: {res[0][0]:.3f} s, same samples: {same}", flush=True)

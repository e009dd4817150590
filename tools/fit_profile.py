"""wall clock and Python profile of GP.fit (design stage + lock-step multi-start + slice sampling) at a size
gpyreg's users work at"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpyreg_amd as gpr

N, D = int(sys.argv[1]), int(sys.argv[2])
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 10
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


gp = make()
gp.fit(X=X, y=y, options={"n_samples": ns})  # warm-up (library load, first allocations)
gp = make()
np.random.seed(4)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
gp.fit(X=X, y=y, options={"n_samples": ns})
pr.disable()
print(f"fit N={N} D={D} n_samples={ns}: {time.perf_counter() - t0:.3f} s", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)

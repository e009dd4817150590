import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time
import numpy as np
import bench
N = int(sys.argv[1]); grad = int(sys.argv[2])
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
X, y, hyp = bench.synthetic_problem(2, 1)
gp = bench.make_gp(2, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
for _ in range(3): gp.nll_batch(hyp, bool(grad))
t0 = time.perf_counter()
for _ in range(20): gp.nll_batch(hyp, bool(grad))
print("N", N, "grad", grad, "ms/call", (time.perf_counter() - t0) / 20 * 1e3)
from gpyreg_amd import _lib
print("device timing (total, factor) ms:", _lib.context().last_timing())

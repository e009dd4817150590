"""per-call latency of single NLL / NLL+grad evaluations at small N: device vs CPU oracle."""
import time
import numpy as np
import bench
from oracle import gp_oracle as orc

for N in (50, 200, 500, 1000, 2000):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, 1)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    model = dict(kernel="se", degree=0, mean="const", noise=(1, 0, 0))
    for grad in (False, True):
        gp.nll_batch(hyp, grad)
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            gp._GP__compute_nlZ(hyp[0], grad, False)
        tg = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        creps = 3 if N > 500 else 10
        for _ in range(creps):
            orc.core(model, hyp[0], X, y, None, 1, 1 if grad else 0)
        tc = (time.perf_counter() - t0) / creps
        print(f"N={N:5d} grad={int(grad)}: device {tg*1e3:7.3f} ms   CPU oracle {tc*1e3:8.3f} ms   x{tc/tg:6.1f}", flush=True)

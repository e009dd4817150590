"""per-call latency of single NLL / NLL+grad evaluations at small N: device vs CPU oracle.
All device timings are taken first: the oracle's BLAS worker threads keep spinning for a while
after a call and would pollute the host-side launch latency of the next device measurement."""
import time
import numpy as np
import bench
from oracle import gp_oracle as orc

SIZES = (50, 200, 500, 1000, 2000)
dev = {}
prob = {}
for N in SIZES:
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, 1)
    prob[N] = (X, y, hyp)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    for grad in (False, True):
        for _ in range(5):
            gp.nll_batch(hyp, grad)
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            gp._GP__compute_nlZ(hyp[0], grad, False)
        dev[N, grad] = (time.perf_counter() - t0) / reps
model = dict(kernel="se", degree=0, mean="const", noise=(1, 0, 0))
for N in SIZES:
    X, y, hyp = prob[N]
    for grad in (False, True):
        creps = 3 if N > 500 else 10
        orc.core(model, hyp[0], X, y, None, 1, 1 if grad else 0)
        t0 = time.perf_counter()
        for _ in range(creps):
            orc.core(model, hyp[0], X, y, None, 1, 1 if grad else 0)
        tc = (time.perf_counter() - t0) / creps
        tg = dev[N, grad]
        print(f"N={N:5d} grad={int(grad)}: device {tg*1e3:7.3f} ms   CPU oracle {tc*1e3:8.3f} ms   x{tc/tg:6.1f}", flush=True)

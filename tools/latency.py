"""per-call latency of single NLL / NLL+grad evaluations at small N on the device.
(The CPU figures quoted beside these in DESIGN.md come from tests/test_gpu_latency_report.py,
which is allowed to run the oracle; tools never import it.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import bench

for N in (50, 200, 500, 1000, 2000, 4096):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, 1)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    for grad in (False, True):
        for _ in range(5):
            gp.nll_batch(hyp, grad)
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            gp._GP__compute_nlZ(hyp[0], grad, False)
        tg = (time.perf_counter() - t0) / reps
        print(f"N={N:5d} grad={int(grad)}: device {tg*1e3:7.3f} ms per evaluation", flush=True)

# GP.update (posteriors of S hyperparameter samples) and GP.predict at a few query points: what an acquisition function
# (PyVBMC's, the reference's plot) calls in a loop -- gaussian_process.py:870-884, :1663-1816
for N, S in ((50, 1), (200, 1), (200, 8), (1000, 1), (1000, 8), (2000, 8)):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp)
    for _ in range(3):
        gp.update(hyp=hyp)
    t0 = time.perf_counter()
    for _ in range(20):
        gp.update(hyp=hyp)
    tu = (time.perf_counter() - t0) / 20
    line = f"N={N:5d} S={S:2d}: update(hyp) {tu*1e3:7.3f} ms;  predict"
    for M in (1, 10, 100, 1000):
        xs = np.random.default_rng(M).uniform(-3, 3, (M, X.shape[1]))
        for _ in range(3):
            gp.predict(xs)
        t0 = time.perf_counter()
        for _ in range(30):
            gp.predict(xs)
        line += f"  M={M}: {(time.perf_counter() - t0) / 30 * 1e3:6.3f} ms"
    print(line, flush=True)

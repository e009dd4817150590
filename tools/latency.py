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

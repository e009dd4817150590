"""NLL-only and NLL+grad wall clock per call against the batch size at small N (the speculative sampler's regime)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
GRAD = len(sys.argv) > 1 and sys.argv[1] == "grad"

for N in (300, 1000, 2000):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, 16)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    line = f"N={N:5d}:"
    for S in (1, 2, 4, 6, 8, 16):
        for _ in range(3):
            gp.nll_batch(hyp[:S], GRAD)
        t0 = time.perf_counter()
        for _ in range(20):
            gp.nll_batch(hyp[:S], GRAD)
        line += f"  S={S}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms"
    print(line, flush=True)

"""posterior + predict throughput on the bench workloads (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import bench

for cfg, S, M in [(3, 16, 1000), (3, 16, 100), (2, 1, 1000), (5, 16, 1000)]:
    X, y, hyp = bench.synthetic_problem(cfg, S)
    gp = bench.make_gp(cfg, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    t0 = time.perf_counter(); gp.update(hyp=hyp); t_up0 = time.perf_counter() - t0
    t0 = time.perf_counter(); gp.update(hyp=hyp); t_up = time.perf_counter() - t0
    xs = np.random.default_rng(1).uniform(-3, 3, (M, X.shape[1]))
    gp.predict(xs)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        mu, s2 = gp.predict(xs)
    t_pr = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter(); _, C = gp.predict_full(xs[:256]); t_pf = time.perf_counter() - t0
    print(f"cfg{cfg} N={X.shape[0]} S={S}: update(posteriors) {t_up*1e3:.1f} ms ({S/t_up:.0f} posteriors/s; first call {t_up0*1e3:.1f} ms); "
          f"predict M={M}: {t_pr*1e3:.2f} ms = {M*S/t_pr/1e6:.2f} M point-samples/s; predict_full M=256: {t_pf*1e3:.1f} ms", flush=True)

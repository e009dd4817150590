"""fp32-vs-fp64 accuracy of NLL and gradient on the bench workloads (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time
import numpy as np
import bench

for cfg, Ns in [(4, [2048, 8192, 16384]), (3, [4096])]:
    for N in Ns:
        bench.CONFIGS[cfg] = dict(bench.CONFIGS[cfg], N=N)
        X, y, hyp = bench.synthetic_problem(cfg, 1)
        out = {}
        for dt in ("f64", "f32"):
            gp = bench.make_gp(cfg, dt)
            gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
            t0 = time.perf_counter()
            out[dt] = gp.nll_batch(hyp, compute_grad=True)
            out[dt + "_t"] = time.perf_counter() - t0
        (n64, d64), (n32, d32) = out["f64"], out["f32"]
        en = abs(n32[0] - n64[0]) / abs(n64[0])
        ed = np.abs(d32[0] - d64[0]) / np.maximum(np.abs(d64[0]), np.abs(d64[0]).max())
        print(f"cfg{cfg} N={N}: nlZ64={n64[0]:.6f} nlZ32={n32[0]:.6f} rel={en:.2e}  grad max rel={ed.max():.2e} "
              f"(|g|max={np.abs(d64[0]).max():.3g})  t64={out['f64_t']*1e3:.1f}ms t32={out['f32_t']*1e3:.1f}ms", flush=True)

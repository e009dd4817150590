"""a few single evaluations at N = 100 for a kernel trace (rocprofv3 --kernel-trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
N = int(os.environ.get("SMALL_N", "100")); S = int(os.environ.get("SMALL_S", "1"))
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
X, y, hyp = bench.synthetic_problem(2, S)
gp = bench.make_gp(2, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
for g in ((False,) if os.environ.get("SMALL_NLL_ONLY") else (False, True)):
    for _ in range(6):
        gp.nll_batch(hyp, g)

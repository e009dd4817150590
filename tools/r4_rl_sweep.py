"""The right-looking NLL-only plan (plan.h: potrf_rl, opt-in) with larger panels and look-ahead limits, against the blocked
recursion (the default): N = 4096, S = 1 .. 64, ms per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

ctx = _lib.context()
for S in (1, 2, 4, 8, 16, 64):
    X, y, hyp = bench.synthetic_problem(3, S)
    gp = bench.make_gp(3, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    row = []
    ref = None
    for panel, ahead in ((0, 8), (512, 8), (512, 64), (1024, 8), (1024, 64), (2048, 64)):
        ctx.set_option("rl_panel", panel)
        ctx.set_option("rl_ahead_max", ahead)
        for _ in range(3):
            v, _ = gp.nll_batch(hyp, False)
        reps = 10 if S <= 16 else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            v, _ = gp.nll_batch(hyp, False)
        t = (time.perf_counter() - t0) / reps
        if ref is None:
            ref = v
        row.append("panel=%d/ahead<=%d: %.3f ms (max rel diff %.1e)" % (panel, ahead, t * 1e3, float(np.max(np.abs(v - ref) / np.abs(ref)))))
    ctx.set_option("rl_panel", 0)
    ctx.set_option("rl_ahead_max", 8)
    print("N=4096 S=%d NLL-only: " % S + " | ".join(row), flush=True)

"""NLL-only evaluations per second against the size of the largest inverted diagonal block (plan.h: potrf_nll).
usage: python tools/nll_block_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

ctx = _lib.context(0)
for cfg, N, S in ():
    bench.CONFIGS[cfg] = dict(bench.CONFIGS[cfg], N=N)
    X, y, hyp = bench.synthetic_problem(cfg, S)
    gp = bench.make_gp(cfg, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    for blk in (0, 128, 256, 512, 1024, 2048, 4096):
        if blk >= N and blk != 0 and blk > 1024:
            continue
        ctx.set_option("nll_block", blk)
        for _ in range(3):
            gp.nll_batch(hyp, False)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            gp.nll_batch(hyp, False)
        dt = (time.perf_counter() - t0) / reps
        print(f"cfg{cfg} N={N} S={S} nll_block={blk:5d}: {dt*1e3:8.3f} ms per batch, {S/dt:9.1f} evals/s, "
              f"{S*N**3/3/dt/1e12:6.2f} TFLOP/s algorithmic", flush=True)
ctx.set_option("nll_block", -1)
# right-looking panels (plan.h: potrf_rl), with and without the look-ahead
for cfg, N, S in ((3, 4096, 1), (3, 4096, 4), (3, 4096, 16), (3, 4096, 64), (5, 8192, 1), (5, 8192, 8), (5, 8192, 24)):
    bench.CONFIGS[cfg] = dict(bench.CONFIGS[cfg], N=N)
    X, y, hyp = bench.synthetic_problem(cfg, S)
    gp = bench.make_gp(cfg, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    for panel, ahead in ((0, 0), (512, 1000), (512, 0), (1024, 1000), (1024, 0)):
        ctx.set_option("rl_panel", panel)
        ctx.set_option("rl_ahead_max", ahead)
        for _ in range(3):
            got, _ = gp.nll_batch(hyp, False)
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            gp.nll_batch(hyp, False)
        dt = (time.perf_counter() - t0) / reps
        print(f"cfg{cfg} N={N} S={S} rl_panel={panel:5d} lookahead={'on ' if ahead else 'off'}: {dt*1e3:8.3f} ms per batch, {S/dt:9.1f} evals/s, "
              f"{S*N**3/3/dt/1e12:6.2f} TFLOP/s algorithmic", flush=True)
ctx.set_option("rl_panel", 0)
ctx.set_option("rl_ahead_max", 8)

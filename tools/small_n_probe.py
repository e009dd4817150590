"""Where a small evaluation spends its time (N <= 256: the regime of fit's 1024-point design and of PyVBMC's GPs):
wall clock of GP.nll_batch, of the bare C call underneath it, and the device section by hipEvents."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib
from gpyreg_amd.gaussian_process import _DTYPES

def timeit(f, reps):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps

for N, S in [(50, 1), (100, 1), (128, 1), (200, 1), (256, 1), (50, 1024), (100, 1024), (128, 1024), (200, 1024), (256, 1024), (500, 1024)]:
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    ctx = _lib.context()
    for grad in (False, True):
        reps = 200 if S == 1 else 5
        t_api = timeit(lambda: gp.nll_batch(hyp, grad), reps)
        # (the device section by hipEvents: one-leaf evaluations record them on request only -- each is a barrier packet)
        ctx.set_option("small_timing", 1)
        gp.nll_batch(hyp, grad)
        gp.nll_batch(hyp, grad)
        dev = ctx.last_timing()[0]
        ctx.set_option("small_timing", 0)
        pv = gp._plugin_values(hyp, grad)
        kid, deg = gp._kid()
        cov_N = gp._counts()[0]
        t_c = timeit(lambda: ctx.nll_batch(kid, deg, _DTYPES[gp.dtype], hyp[:, :cov_N], pv["m"], pv["sn2"], pv["vec"], grad, pv["dm"], pv["dsn2"]), reps)
        print(f"N={N:4d} S={S:5d} {'NLL+grad' if grad else 'NLL     '}: GP.nll_batch {t_api*1e6:9.1f} us   C call {t_c*1e6:9.1f} us   device {dev*1e3:9.1f} us"
              f"   ({S/t_api:9.0f} evals/s)", flush=True)
print("one-leaf calls completed by the polled word:", ctx.get_option("small_polled"), " by a stream synchronisation:", ctx.get_option("small_synced"))

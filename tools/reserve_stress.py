"""Stress of the CU-reserving launches (deferred inverse products and the split covariance build; common.h:
cu_reserve_bail): hundreds of pipelines with the deferred schedule forced on, the reservation level changing from run to
run (including "every CU reserved", where one surviving block does all the work of a launch), `check_queues` verifying
after every pipeline that each tile queue was handed out completely, every result compared bit for bit with the in-order
schedule.  usage: python tools/reserve_stress.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

ctx = _lib.context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(7)
cfg = dict(bench.CONFIGS[3])
bad = 0
t0 = time.time()
print(ctx.device_info(), flush=True)
ctx.set_option("check_queues", 1)
try:
    for N, S in ((2304, 6), (4096, 4), (2048, 16)):
        bench.CONFIGS[3] = dict(cfg, N=N)
        X, y, hyp = bench.synthetic_problem(3, S)
        gp = bench.make_gp(3, "f64")
        gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
        for it in range(iters):
            h = hyp + 0.2 * rng.standard_normal(hyp.shape)
            ctx.set_option("defer_min", 0)
            ref = gp.nll_batch(h, compute_grad=True)
            rsv = [2, 4, 8, 12, 32][it % 5] if it % 7 else 32
            ctx.set_option("defer_min", 512 if it % 2 else -1)
            ctx.set_option("defer_reserve", rsv)
            got = gp.nll_batch(h, compute_grad=True)
            if not (np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])):
                bad += 1
                print("MISMATCH", N, S, it, rsv, flush=True)
        print(f"N={N} S={S}: {iters} deferred pipelines compared with the in-order schedule, mismatches so far {bad}, "
              f"{time.time() - t0:.1f} s", flush=True)
finally:
    ctx.set_option("check_queues", 0)
    ctx.set_option("defer_min", -1)
    ctx.set_option("defer_reserve", 8)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

"""single evaluations at small N: wall clock per call against the device section (hipEvents), host phases
(GPC_HOSTTIME) and a Python profile of the call path"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

N = int(sys.argv[1])
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
X, y, hyp = bench.synthetic_problem(2, 1)
gp = bench.make_gp(2, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
for _ in range(10):
    gp._GP__compute_nlZ(hyp[0], True, False)
reps = 200
t0 = time.perf_counter()
dev = 0.0
for _ in range(reps):
    gp._GP__compute_nlZ(hyp[0], True, False)
    dev += _lib.context().last_timing()[0]
t = (time.perf_counter() - t0) / reps
print(f"N={N}: wall {t*1e6:.1f} us per evaluation, device events {dev/reps*1e3:.1f} us", flush=True)
os.environ["GPC_HOSTTIME"] = "1"
gp._GP__compute_nlZ(hyp[0], True, False)
del os.environ["GPC_HOSTTIME"]
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    gp._GP__compute_nlZ(hyp[0], True, False)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)

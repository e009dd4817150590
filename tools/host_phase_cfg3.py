"""cfg3 step on the host side: wall clock of GP.nll_batch against the device section (hipEvents), the library's host
phases (GPC_HOSTTIME) and a Python profile of ten calls.  Round 3: wall 18.94 ms, device 18.77 ms -- 0.17 ms of host
(fill 0.03, upload 0.03, 0.53 ms of launches issued under the device's work)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from gpyreg_amd import _lib
X, y, hyp = bench.synthetic_problem(3, 16)
gp = bench.make_gp(3, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
for _ in range(5): gp.nll_batch(hyp, True)
ctx=_lib.context()
w=[];d=[]
for _ in range(20):
    t0=time.perf_counter(); gp.nll_batch(hyp, True); w.append((time.perf_counter()-t0)*1e3); d.append(ctx.last_timing()[0])
print("wall %.3f ms  device %.3f ms  diff %.3f"%(np.median(w), np.median(d), np.median(w)-np.median(d)))
os.environ["GPC_HOSTTIME"]="1"
gp.nll_batch(hyp, True)
del os.environ["GPC_HOSTTIME"]
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(10): gp.nll_batch(hyp, True)
pr.disable()
st=pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)

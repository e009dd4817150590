"""where does the wall time of nll_batch calls go? (GPC_HOSTTIME phases + wall clock)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys, time
import numpy as np
import bench
from gpyreg_amd import _lib
_lib.context().set_option("small_timing", 1)  # (below N_pad = 2048 the timing events are recorded on request only)

N, S = int(sys.argv[1]), int(sys.argv[2])
grad = len(sys.argv) > 3 and sys.argv[3] == "grad"
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
X, y, hyp = bench.synthetic_problem(2, S)
gp = bench.make_gp(2, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
gp.nll_batch(hyp, grad)
for label, h in (("batch", hyp), ("single", hyp[:1])):
    for _ in range(3):
        gp.nll_batch(h, grad)
    os.environ["GPC_HOSTTIME"] = "1"
    t0 = time.perf_counter()
    gp.nll_batch(h, grad)
    t = (time.perf_counter() - t0)
    del os.environ["GPC_HOSTTIME"]
    print(f"== {label}: {t*1e3:.2f} ms/call, device {_lib.context().last_timing()[0]:.2f} ms", flush=True)

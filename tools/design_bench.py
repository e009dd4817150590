"""throughput of the batched design evaluation (fit's f_min_fill stage): S NLL evaluations at once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import bench
from gpyreg_amd import _lib
_lib.context().set_option("small_timing", 1)  # (below N_pad = 2048 the timing events are recorded on request only)

for N, S in [(200, 1024), (500, 1024), (1000, 1024), (2000, 256)]:
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    gp.nll_batch(hyp, False)
    t0 = time.perf_counter()
    for _ in range(3):
        nlz, _ = gp.nll_batch(hyp, False)
    t = (time.perf_counter() - t0) / 3
    dev = _lib.context().last_timing()[0]
    t0 = time.perf_counter()
    for s in range(8):
        gp.nll_batch(hyp[s:s + 1], False)
    t1 = (time.perf_counter() - t0) / 8
    print(f"N={N} S={S}: batched {t*1e3:.1f} ms ({S/t:.0f} evals/s; device {dev:.1f} ms); one at a time {t1*1e3:.2f} ms/eval ({1/t1:.0f}/s); batching x{t1*S/t:.1f}", flush=True)

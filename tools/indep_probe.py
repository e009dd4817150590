"""Independent pipelines (gpc_set_option("indep", 1)) against the lock-step batch: bits and time per call.
usage: python tools/indep_probe.py N grad S [S ...]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

N, grad = int(sys.argv[1]), bool(int(sys.argv[2]))
Ss = [int(v) for v in sys.argv[3:]] or [2, 3, 4]
ctx = _lib.context(0)
bench.CONFIGS[3] = dict(bench.CONFIGS[3], N=N)
for k, v in (kv.split("=") for kv in os.environ.get("INDEP_OPTS", "").split() if kv):
    ctx.set_option(k, int(v))
    print("option", k, "=", v)
for S in Ss:
    X, y, hyp = bench.synthetic_problem(3, S)
    gp = bench.make_gp(3, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)

    def timed(reps=9):
        out = gp.nll_batch(hyp, compute_grad=grad)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = gp.nll_batch(hyp, compute_grad=grad)
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, min(ts), float(np.median(ts))

    ctx.set_option("indep", 0)
    ref, a0, b0 = timed()
    ctx.set_option("indep", 1)
    got, a1, b1 = timed()
    ctx.set_option("indep", 0)
    ref2, a2, b2 = timed()
    same = np.array_equal(ref[0], got[0]) and (not grad or np.array_equal(ref[1], got[1], equal_nan=True))
    print(f"N={N} S={S} grad={int(grad)}: lock-step min {a0:7.3f} med {b0:7.3f} | independent min {a1:7.3f} med {b1:7.3f} | lock-step again "
          f"min {a2:7.3f} med {b2:7.3f} ms   identical: {same}", flush=True)

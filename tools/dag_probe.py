"""Dataflow graph (option "dag") against the stream-ordered schedule: bits and time per call.
usage: python tools/dag_probe.py N S [grad=1] [dtype=f64] [reps=5] [key=value ...options]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

N, S = int(sys.argv[1]), int(sys.argv[2])
grad = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
dtype = sys.argv[4] if len(sys.argv) > 4 else "f64"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
opts = dict(kv.split("=") for kv in sys.argv[6:])
ctx = _lib.context(0)
bench.CONFIGS[3] = dict(bench.CONFIGS[3], N=N)
X, y, hyp = bench.synthetic_problem(3, S)
gp = bench.make_gp(3, dtype)
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)


def timed(tag):
    out = gp.nll_batch(hyp, compute_grad=grad)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = gp.nll_batch(hyp, compute_grad=grad)
        ts.append((time.perf_counter() - t0) * 1e3)
    dev = ctx.last_timing()
    print(f"{tag:28s} wall ms: min {min(ts):8.3f} median {np.median(ts):8.3f}   device section {dev[1]:8.3f} ms", flush=True)
    return out


ctx.set_option("dag", 0)
ref = timed("stream-ordered")
for k, v in opts.items():
    ctx.set_option(k, int(v))
ctx.set_option("dag", 1)
r0, a0 = ctx.get_option("dag_runs"), ctx.get_option("dag_aborts")
got = timed("dataflow graph " + " ".join(f"{k}={v}" for k, v in opts.items()))
again = gp.nll_batch(hyp, compute_grad=grad)
print("graph runs", ctx.get_option("dag_runs") - r0, "aborts", ctx.get_option("dag_aborts") - a0)
print("nlZ identical:", np.array_equal(ref[0], got[0]), " max rel diff %.2e" % np.max(np.abs(ref[0] - got[0]) / np.abs(ref[0])),
      " graph deterministic:", np.array_equal(got[0], again[0]))
if grad:
    print("dnlZ identical:", np.array_equal(ref[1], got[1]), " max abs diff %.2e" % np.max(np.abs(ref[1] - got[1])))

"""128 x 64 tiles for the launches below the 128-tile threshold (gpc_set_option("rect_min", n)): bits and time per call.
usage: python tools/rect_probe.py N S grad [rect_min ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

N, S, grad = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
mode = int(os.environ.get("RECT_MODE", "0"))
mins = [int(v) for v in sys.argv[4:]] or [64, 128, 256, 512]
ctx = _lib.context(0)
bench.CONFIGS[3] = dict(bench.CONFIGS[3], N=N)
X, y, hyp = bench.synthetic_problem(3, S)
gp = bench.make_gp(3, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)


def timed(reps=7):
    out = gp.nll_batch(hyp, compute_grad=grad)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = gp.nll_batch(hyp, compute_grad=grad)
        ts.append((time.perf_counter() - t0) * 1e3)
    return out, min(ts), float(np.median(ts))


ctx.set_option("rect_mode", mode)
ctx.set_option("rect_min", 0)
ref, a, b = timed()
print(f"(rect_mode {mode}: {'128 x 128 tiles of eight waves' if mode else '128 x 64 tiles'})")
print(f"N={N} S={S} grad={int(grad)}  64-tiles only      : min {a:8.3f} median {b:8.3f} ms", flush=True)
for m in mins:
    ctx.set_option("rect_min", m)
    got, a, b = timed()
    same = np.array_equal(ref[0], got[0]) and (not grad or np.array_equal(ref[1], got[1], equal_nan=True))
    print(f"                     rect_min = {m:5d}   : min {a:8.3f} median {b:8.3f} ms   identical: {same}", flush=True)
ctx.set_option("rect_min", 0)
ref2, a, b = timed()
print(f"                     64-tiles only again: min {a:8.3f} median {b:8.3f} ms", flush=True)

"""The size envelope, exercised: one NLL + gradient evaluation and one NLL-only evaluation at N = gpc_max_n(fp64) (97 920 on
a 288 GB MI355X: three slabs of 82.4 GB), SE kernel, D = 5 -- finite results, NLL-only == NLL of NLL + gradient to rounding,
wall clock and the fraction of the fp64 MFMA peak.   usage: python tools/max_n_probe.py [N | 0] [f64 | f32]   (GPU box; ~1-2 minutes; N = 0: gpc_max_n of the dtype)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

lib = _lib.load()
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
peak = 78.6e12 if dtype == "f64" else 157.3e12
N = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else lib.gpc_max_n(_lib.F64 if dtype == "f64" else _lib.F32)
print(f"gpc_max_n: fp64 {lib.gpc_max_n(_lib.F64)}, fp32 {lib.gpc_max_n(_lib.F32)}; probing N = {N}", flush=True)
bench.CONFIGS[6] = dict(bench.CONFIGS[6], N=N)
X, y, hyp = bench.synthetic_problem(6, 1)
gp = bench.make_gp(6, dtype)
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
t0 = time.perf_counter()
nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
t1 = time.perf_counter()
n0, _ = gp.nll_batch(hyp, compute_grad=False)
t2 = time.perf_counter()
print(f"NLL + gradient: {t1 - t0:.2f} s = {N ** 3 / (t1 - t0) / 1e12:.1f} TFLOP/s ({N ** 3 / (t1 - t0) / peak:.3f} of {peak / 1e12}, first call: allocation included); "
      f"NLL only: {t2 - t1:.2f} s = {N ** 3 / 3 / (t2 - t1) / 1e12:.1f} TFLOP/s")
print("nlZ", nlz[0], "NLL-only", n0[0], "rel diff", abs(nlz[0] - n0[0]) / abs(nlz[0]), "gradient finite:", bool(np.isfinite(dnlz).all()))
assert np.isfinite(nlz).all() and np.isfinite(dnlz).all() and abs(nlz[0] - n0[0]) <= (1e-10 if dtype == "f64" else 1e-3) * abs(nlz[0])
t0 = time.perf_counter()
nlz2, _ = gp.nll_batch(hyp, compute_grad=True)
t1 = time.perf_counter()
assert nlz2[0] == nlz[0]
print(f"NLL + gradient again (workspace in place): {t1 - t0:.2f} s = {N ** 3 / (t1 - t0) / peak:.3f} of the {dtype} MFMA peak by the wall clock")

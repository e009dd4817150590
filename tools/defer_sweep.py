"""NLL+grad batch time for several (N, S) shapes with and without the deferred inverse products (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

ctx = _lib.context(0)
shapes = [(2048, 16), (2048, 32), (3000, 16), (4096, 4), (4096, 8), (4096, 16), (4096, 32), (6000, 16), (8192, 16)]
for N, S in shapes:
    bench.CONFIGS[3] = dict(bench.CONFIGS[3], N=N)
    X, y, hyp = bench.synthetic_problem(3, S)
    gp = bench.make_gp(3, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    res = []
    for groups, dmin, rsv in [(2, 0, 0), (1, 0, 0), (1, -2, 8), (1, -2, 4), (1, -4, 8)]:
        ctx.set_option("groups", groups)
        npad = ((N + 127) // 128) * 128
        ctx.set_option("defer_min", 0 if dmin == 0 else max(256, (npad // -dmin) // 128 * 128))
        ctx.set_option("defer_reserve", rsv)
        for _ in range(2):
            gp.nll_batch(hyp, True)
        t0 = time.perf_counter(); reps = 5 if N <= 4096 else 2
        for _ in range(reps):
            gp.nll_batch(hyp, True)
        res.append((time.perf_counter() - t0) / reps * 1e3)
    print(f"N={N:5d} S={S:3d}: groups2 {res[0]:8.2f}  groups1 {res[1]:8.2f}  defer(n/2,r8) {res[2]:8.2f}  defer(n/2,r4) {res[3]:8.2f}  defer(n/4,r8) {res[4]:8.2f} ms", flush=True)

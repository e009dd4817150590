"""Race hunt for the pipelined leaf: thousands of factorizations, single and batched (many leaf blocks in flight, other
kernels around them), compared bit for bit with the phase-ordered leaf."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

ctx = _lib.context(0)
rng = np.random.default_rng(123)
bad = 0
t0 = time.time()
n_single = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
SCALE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for it in range(n_single):
    n = int(rng.choice([16, 100, 128, 129, 200, 256, 300, 384]))
    X = rng.uniform(-3, 3, (n, 2))
    d = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    A = np.exp(-0.5 * d / rng.uniform(0.5, 4.0)) / rng.uniform(0.005, 0.5) + np.eye(n)
    out = []
    for leaf in (3, 5):
        ctx.set_option("leaf", leaf)
        out.append(ctx.debug_factor(A, want_inv=False))
    (L3, W3, _, ld3, i3), (L5, W5, _, ld5, i5) = out
    if not (np.array_equal(L3, L5) and np.array_equal(W3, W5) and ld3 == ld5 and i3 == i5):
        bad += 1
        print("MISMATCH single", it, n, flush=True)
print(f"{n_single} single factorizations: {bad} mismatches, {time.time()-t0:.1f} s", flush=True)
cfg = dict(bench.CONFIGS[3])
for N, S, reps in ((300, 256, 40 * SCALE), (1000, 64, 30 * SCALE), (2304, 16, 15 * SCALE)):
    bench.CONFIGS[3] = dict(cfg, N=N)
    X, y, hyp = bench.synthetic_problem(3, S)
    for rep in range(reps):
        h = hyp + 0.2 * rng.standard_normal(hyp.shape)
        res = []
        for leaf in (3, 5):
            ctx.set_option("leaf", leaf)
            gp = bench.make_gp(3, "f64")
            gp.update(X_new=X, y_new=y, hyp=h[:1], compute_posterior=False)
            res.append(gp.nll_batch(h, compute_grad=True))
        if not (np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])):
            bad += 1
            print("MISMATCH batch", N, S, rep, flush=True)
    print(f"N={N} S={S}: {reps} batched evaluations compared, total mismatches {bad}, {time.time()-t0:.1f} s", flush=True)
ctx.set_option("leaf", 5)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

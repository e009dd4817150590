"""Soak: a few minutes of randomly mixed calls through the public API -- fits of random sizes and kernels, batched
NLL / gradients, posteriors, predictions, one-point updates (device and user-defined kernels), singular systems that
take the jitter-retry path, forced leaf time-outs (error path) -- checking that every call either succeeds with
finite results or raises the documented exception, that the library keeps working after errors, and that free device
memory does not drift.  usage: python tools/soak.py [seconds]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import gpyreg_amd as gpr
from gpyreg_amd import _lib
from test_gpu_user_kernel import PySquaredExponential

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2026)
ctx = _lib.context(0)


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def make(D, kind, user=False):
    cov = {0: gpr.covariance_functions.SquaredExponential, 1: lambda: gpr.covariance_functions.Matern(int(rng.choice([1, 3, 5]))),
           2: gpr.covariance_functions.RationalQuadraticARD,
           3: gpr.isotropic_covariance_functions.SquaredExponentialIsotropic}[kind]()
    if user:
        cov = PySquaredExponential()
    mean = [gpr.mean_functions.ZeroMean, gpr.mean_functions.ConstantMean, gpr.mean_functions.NegativeQuadratic][int(rng.integers(3))]()
    return gpr.GP(D, cov, mean, gpr.noise_functions.GaussianNoise(constant_add=True), dtype="f64" if rng.random() < 0.8 else "f32")


def hyp_for(gp, S, D, log_noise):
    cov_N, noise_N, mean_N = gp._counts()
    h = 0.3 * rng.standard_normal((S, cov_N + noise_N + mean_N))
    h[:, :min(D, cov_N)] += np.log(1.5)
    h[:, cov_N] = log_noise
    if mean_N > 1:
        h[:, cov_N + noise_N + 1 + D:] = np.log(3.0)
    return h


t0 = time.time()
stats = dict(calls=0, linalg=0, runtime=0, appended=0, retried=0)
base_free = None
it = 0
while time.time() - t0 < budget:
    it += 1
    D = int(rng.integers(1, 5))
    N = int(rng.choice([5, 40, 127, 128, 129, 300, 700, 1100]))
    S = int(rng.choice([1, 2, 5, 9]))
    user = rng.random() < 0.15 and N <= 300
    gp = make(D, int(rng.integers(4)), user)
    X = rng.uniform(-3, 3, (N, D))
    singular = rng.random() < 0.15
    if singular and N >= 10:
        X[N // 2:] = X[:N - N // 2] + 1e-9 * rng.standard_normal((N - N // 2, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    hyp = hyp_for(gp, S, D, np.log(1e-7) if singular else np.log(0.1))
    try:
        gp.update(X_new=X, y_new=y, hyp=hyp)
        stats["retried"] += sum(p.sn2_mult != 1 for p in gp.posteriors)
        nl, dn = gp.nll_batch(hyp, compute_grad=True)
        n0, _ = gp.nll_batch(hyp)
        ok = ~np.isnan(dn)  # Matern-1 length-scale gradients are NaN on purpose
        assert np.isfinite(nl).all() and np.isfinite(dn[ok]).all() and np.isfinite(n0).all()
        mu, s2 = gp.predict(X[: min(N, 17)] + 0.01, separate_samples=True)
        assert np.isfinite(mu).all() and np.isfinite(s2).all() and (s2 >= 0).all()
        for _ in range(int(rng.integers(0, 3))):
            xn = rng.uniform(-3, 3, (1, D))
            gp.update(X_new=xn, y_new=np.sin(xn.sum(1, keepdims=True)))
            stats["appended"] += 1
        mu, s2 = gp.predict(X[: min(N, 9)] - 0.02)
        assert np.isfinite(mu).all() and np.isfinite(s2).all()
        if it % 11 == 0:  # the error path: a forced leaf time-out must raise, and the next call must work
            ctx.set_option("leaf_fault", 1)
            try:
                gp.nll_batch(hyp)
                raise AssertionError("a leaf time-out did not raise")
            except RuntimeError:
                stats["runtime"] += 1
            finally:
                ctx.set_option("leaf_fault", 0)
            assert np.isfinite(gp.nll_batch(hyp)[0]).all()
    except np.linalg.LinAlgError:
        stats["linalg"] += 1  # documented: still not positive definite after ten jitter levels
    stats["calls"] += 1
    del gp
    if it % 25 == 0:
        gc.collect()
        f = free_mib()
        base_free = base_free if base_free is not None else f
        print(f"{time.time() - t0:6.1f} s  {stats}  free {f:.0f} MiB (first reading {base_free:.0f})", flush=True)
gc.collect()
f = free_mib()
print("done", stats, f"free {f:.0f} MiB, first reading {base_free:.0f}")
assert base_free is None or f > base_free - 512, "device memory drifted"

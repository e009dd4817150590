"""User-defined covariance objects through the plugin protocol (reference
covariance_functions.py:9-20 AbstractKernel, gaussian_process.py:2388-2390: the reference calls
whatever object it was given).  A Python kernel's own compute() supplies K and dK; the
factorization, solves and gradient contraction run on the device (gpc_nll_batch_K,
gpc_posterior_batch_K, gpc_predict_K).  Checked against the reference's golden values (a Python
re-implementation of SE must reproduce the built-in SE fixture) and against the oracle."""

import numpy as np
import pytest
from scipy.spatial.distance import cdist

from conftest import parse_core_name
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


class PySquaredExponential:
    """SE-ARD written against the reference's AbstractKernel protocol, NumPy only."""

    def hyperparameter_count(self, D):
        return D + 1

    def hyperparameter_info(self, D):
        return [("covariance_log_lengthscale", D), ("covariance_log_outputscale", 1)]

    def get_bounds_info(self, X, y):
        D = X.shape[1]
        return {"LB": np.full(D + 1, -10.0), "UB": np.full(D + 1, 10.0), "PLB": np.full(D + 1, -2.0),
                "PUB": np.full(D + 1, 2.0), "x0": np.zeros(D + 1)}

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        N, D = X.shape
        if hyp.size != D + 1:
            raise ValueError(f"Expected {D + 1} covariance function hyperparameters, {hyp.size} passed instead.")
        ell, sf2 = np.exp(hyp[:D]), np.exp(2 * hyp[D])
        if compute_diag:
            return sf2 * np.ones((N, 1))
        Xs = X / ell
        Ys = Xs if X_star is None else X_star / ell
        K = sf2 * np.exp(-0.5 * cdist(Xs, Ys, "sqeuclidean"))
        if not compute_grad:
            return K
        dK = np.empty((N, N, D + 1))
        for d in range(D):
            dK[:, :, d] = K * (Xs[:, d:d + 1] - Xs[:, d:d + 1].T) ** 2
        dK[:, :, D] = 2 * K
        return K, dK


class PyPeriodicPlusLinear:
    """A kernel the library has no device code for: sf2 exp(-2 sin^2(pi |x-x'| / p) / ell^2) + sl2 x.x'."""

    def hyperparameter_count(self, D):
        return 4

    def hyperparameter_info(self, D):
        return [("covariance_log_lengthscale", 1), ("covariance_log_period", 1), ("covariance_log_outputscale", 1),
                ("covariance_log_linear", 1)]

    def get_bounds_info(self, X, y):
        return {"LB": np.full(4, -5.0), "UB": np.full(4, 5.0), "PLB": np.full(4, -1.0), "PUB": np.full(4, 1.0),
                "x0": np.zeros(4)}

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        ell, per, sf2, sl2 = np.exp(hyp[0]), np.exp(hyp[1]), np.exp(2 * hyp[2]), np.exp(2 * hyp[3])
        if compute_diag:
            return (sf2 + sl2 * np.sum(X * X, 1))[:, None]
        Y = X if X_star is None else X_star
        r = cdist(X, Y)
        s = np.sin(np.pi * r / per)
        Kp = sf2 * np.exp(-2 * s**2 / ell**2)
        Kl = sl2 * (X @ Y.T)
        if not compute_grad:
            return Kp + Kl
        N = X.shape[0]
        dK = np.empty((N, N, 4))
        dK[:, :, 0] = Kp * (4 * s**2 / ell**2)
        dK[:, :, 1] = Kp * (4 * s * np.cos(np.pi * r / per) * np.pi * r / per / ell**2)
        dK[:, :, 2] = 2 * Kp
        dK[:, :, 3] = 2 * Kl
        return Kp + Kl, dK


def _mk(cov, D, mean="const", noise=(1, 0, 0), dtype="f64"):
    import gpyreg_amd as gpr

    m = {"zero": gpr.mean_functions.ZeroMean, "const": gpr.mean_functions.ConstantMean,
         "negquad": gpr.mean_functions.NegativeQuadratic}[mean]()
    n = gpr.noise_functions.GaussianNoise(constant_add=noise[0] == 1, user_provided_add=noise[1] >= 1,
                                          scale_user_provided=noise[1] == 2,
                                          rectified_linear_output_dependent_add=noise[2] == 1)
    return gpr.GP(D, cov, m, n, dtype=dtype)


def test_python_se_kernel_reproduces_the_reference_goldens(core_golden):
    """Every SE fixture (all means, scalar / per-point / output-dependent noise, low-noise branch)
    through a PYTHON covariance object: nlZ, gradient, posterior and predictions at 1e-8."""
    g = core_golden
    n = 0
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        if model["kernel"] != "se" or flavour != "plain":
            continue
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        gp = _mk(PySquaredExponential(), D, model["mean"], model["noise"])
        assert not gp._builtin
        gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp)
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
        n0, _ = gp.nll_batch(hyp, compute_grad=False)
        for s in range(hyp.shape[0]):
            rn, rd = g[tag + "_nlZ"][s], g[tag + "_dnlZ"][s]
            assert abs(nlz[s] - rn) <= 1e-8 * max(1.0, abs(rn)), (name, s)
            assert abs(n0[s] - rn) <= 1e-8 * max(1.0, abs(rn)), (name, s)
            assert (np.abs(dnlz[s] - rd) <= 1e-8 * np.maximum(np.abs(rd), np.abs(rd).max())).all(), (name, s)
            ra = g[tag + "_alpha"][s]
            assert np.abs(gp.posteriors[s].alpha[:, 0] - ra).max() <= 1e-8 * np.abs(ra).max(), (name, s)
            assert gp.posteriors[s].sn2_mult == g[tag + "_sn2_mult"][s]
        xs, ys = g[tag + "_xs"], g[tag + "_ys"]
        s2s = 0.02 * np.ones((xs.shape[0], 1)) if s2 is not None else None
        mu, v = gp.predict(xs, ys, s2s, add_noise=False, separate_samples=True)
        assert np.abs(mu - g[tag + "_mu_sep"]).max() <= 1e-8 * max(1.0, np.abs(g[tag + "_mu_sep"]).max()), name
        assert np.abs(v - g[tag + "_s2_sep"]).max() <= 1e-7 * max(1.0, np.abs(g[tag + "_s2_sep"]).max()), name
        mu_a, s2_a, lpd_a = gp.predict(xs, ys, s2s, add_noise=True, return_lpd=True)
        assert np.allclose(lpd_a, g[tag + "_lpd_avg"], rtol=1e-6, atol=1e-8), name
        n += 1
    assert n >= 10


def test_kernel_without_device_code_matches_the_oracle():
    """A periodic + linear kernel (no device implementation exists): NLL, gradient, predict and
    predict_full against the oracle evaluating the same Python object, fp64 1e-8 and fp32 1e-3."""
    rng = np.random.default_rng(5)
    N, D, S = 300, 1, 3
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(2 * X) + 0.3 * X + 0.1 * rng.standard_normal((N, 1))
    cov = PyPeriodicPlusLinear()
    hyp = np.array([0.1, np.log(3.0), 0.0, -1.0, np.log(0.1), 0.05]) + 0.1 * rng.standard_normal((S, 6))
    model = dict(kernel=cov, degree=0, mean="const", noise=(1, 0, 0))
    xs = rng.uniform(-3.5, 3.5, (9, D))
    posts = orc.posteriors(model, hyp, X, y, None)
    rmu, rs2 = orc.predict(model, posts, X, y, xs, separate_samples=True)
    for dtype, tol in (("f64", 1e-8), ("f32", 1e-3)):
        gp = _mk(cov, D, dtype=dtype)
        gp.update(X_new=X, y_new=y, hyp=hyp)
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
        for s in range(S):
            rn, rd = orc.core(model, hyp[s], X, y, None, 1, 1)
            assert abs(nlz[s] - rn) <= tol * max(1.0, abs(rn)), (dtype, s)
            assert (np.abs(dnlz[s] - rd) <= tol * np.maximum(np.abs(rd), np.abs(rd).max())).all(), (dtype, s)
        mu, s2 = gp.predict(xs, separate_samples=True)
        assert np.abs(mu - rmu).max() <= tol * max(1.0, np.abs(rmu).max())
        assert np.abs(s2 - rs2).max() <= 10 * tol * max(1.0, np.abs(rs2).max())
        fmu, fcov = gp.predict_full(xs)
        assert np.abs(fmu - rmu).max() <= tol * max(1.0, np.abs(rmu).max())
        assert np.abs(np.stack([np.diag(fcov[:, :, s]) for s in range(S)], 1) - rs2).max() <= 10 * tol * np.abs(rs2).max()
        assert np.allclose(fcov, fcov.transpose(1, 0, 2))
    # through the name-mangled entry point the reference's tests use, and log_likelihood
    a, b = gp._GP__compute_nlZ(hyp[0], True, False)
    assert np.isfinite(a) and np.isfinite(b).all()


def test_user_kernel_jitter_escalation_and_errors(core_golden):
    """A near-singular user K goes through the same x10 escalation as the built-in path (same
    multiplier, same values bit for bit is not required: K comes from NumPy); a dK callback failure
    surfaces as an exception."""
    g = core_golden
    name = [n for n in g["names"] if "jitter_high" in n and n.split("|")[1] == "se"][0]
    tag, model, N, D, flavour = parse_core_name(name)
    X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
    gp = _mk(PySquaredExponential(), D)
    gp.update(X_new=X, y_new=y, hyp=hyp)
    ref = _mk(__import__("gpyreg_amd").covariance_functions.SquaredExponential(), D)
    ref.update(X_new=X, y_new=y, hyp=hyp)
    for a, b in zip(gp.posteriors, ref.posteriors):
        assert a.sn2_mult >= 10 and 0.1 <= a.sn2_mult / b.sn2_mult <= 10
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    assert np.isfinite(nlz).all() and np.isfinite(dnlz).all()
    rng = np.random.default_rng(1)

    class Broken(PySquaredExponential):
        def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
            if compute_grad:
                K, dK = super().compute(hyp, X, compute_grad=True)
                return K, dK[:, :-1, :]  # wrong shape: the plane copy must fail loudly
            return super().compute(hyp, X, X_star, compute_diag)

    bad = _mk(Broken(), D)
    bad.update(X_new=X[:10], y_new=y[:10], hyp=hyp, compute_posterior=False)
    with pytest.raises(Exception):
        bad.nll_batch(hyp, compute_grad=True)

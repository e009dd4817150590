"""gpc_nll_batch_cm -- the stock zero / constant mean crossing the boundary as ONE value per sample (m0[S]; r = y - m0 and,
with scalar noise, the diagonal term are formed on the device from the resident y) -- must return what gpc_nll_batch returns
for m[s][i] = m0[s], dm = 1: the SAME BITS, at every pipeline (one leaf, launch graph, deferred schedule), both precisions,
scalar and per-point noise, with and without gradient, jitter retries included (mean_functions.py:82-131, :210-260;
gaussian_process.py:2371-2400, :2507-2508)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gpyreg_amd import _lib

    return _lib.context(0)


def _both(ctx, gp, hyp, grad):
    """(general entry with broadcast arrays, constant-mean entry) through the bare context"""
    from gpyreg_amd.gaussian_process import _DTYPES

    cov_N, noise_N, mean_N = gp._counts()
    gp._ctx()
    pv = gp._plugin_values(hyp, grad)
    kid, deg = gp._kid()
    a = ctx.nll_batch(kid, deg, _DTYPES[gp.dtype], hyp[:, :cov_N], pv["m"], pv["sn2"], pv["vec"], grad, pv["dm"], pv["dsn2"])
    m0 = hyp[:, cov_N + noise_N] if mean_N == 1 else None
    b = ctx.nll_batch_cm(kid, deg, _DTYPES[gp.dtype], hyp[:, :cov_N], m0, pv["sn2"], pv["vec"], grad, pv["dsn2"])
    return a, b


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N", [60, 128, 300, 1000, 2300])
def test_constant_mean_entry_is_the_general_entry_bit_for_bit(ctx, N, dtype):
    import gpyreg_amd as gpr

    rng = np.random.default_rng(N)
    D, S = 3, 5
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    s2 = 0.01 + 0.02 * rng.uniform(size=(N, 1))
    cases = [(gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True), None),
             (gpr.mean_functions.ZeroMean(), gpr.noise_functions.GaussianNoise(constant_add=True), None),
             (gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True), s2)]
    for mean, noise, s2_ in cases:
        gp = gpr.GP(D, gpr.covariance_functions.Matern(5), mean, noise, dtype=dtype)
        cov_N, noise_N, mean_N = gp._counts()
        hyp = np.concatenate([np.log(1.5) * np.ones(D), [0.0], [np.log(0.2)] * noise_N, [0.3] * mean_N]) \
            + 0.1 * rng.standard_normal((S, cov_N + noise_N + mean_N))
        gp.update(X_new=X, y_new=y, s2_new=s2_, hyp=hyp[:1], compute_posterior=False)
        for grad in (False, True):
            a, b = _both(ctx, gp, hyp, grad)
            assert np.array_equal(a[0], b[0]), (N, dtype, type(mean).__name__, grad)
            if grad:
                assert np.array_equal(a[1], b[1]), (N, dtype, type(mean).__name__)
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
        # and the GP's own methods take the constant-mean entry: same values as the bare call
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
        assert np.array_equal(nlz, b[0]) and np.array_equal(dnlz, b[1])


def test_constant_mean_entry_through_a_jitter_retry(ctx):
    import gpyreg_amd as gpr

    rng = np.random.default_rng(3)
    N, D = 200, 2
    X = rng.uniform(-3, 3, (N, D))
    X[100:] = X[:100]  # duplicated inputs: singular without noise
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    gp = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    hyp = np.array([[np.log(2.0), np.log(2.0), 0.0, np.log(1e-9), 0.2], [np.log(1.0), np.log(1.5), 0.1, np.log(0.1), 0.0]])
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    a, b = _both(ctx, gp, hyp, True)
    assert np.isfinite(b[0]).all() and b[2][0] > 1  # the first sample needed jitter
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])

"""GPU test of the rank-one posterior append (reference
test_gaussian_process.py:387-411: rank-1 updates == full recompute)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(kernel="se"):
    import gpyreg_amd as gpr

    cov = gpr.covariance_functions.SquaredExponential() if kernel == "se" else gpr.covariance_functions.Matern(5)
    return gpr.GP(2, cov, gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True))


def _data(N, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, 2))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    hyp = np.array([[0.1, -0.1, 0.05, np.log(0.2), 0.1], [0.3, 0.2, -0.1, np.log(0.1), -0.2]])
    return X, y, hyp


@pytest.mark.parametrize("N,start,kernel", [(20, 10, "se"), (140, 120, "matern5"), (260, 255, "se")])
def test_rank_one_updates_equal_full_recompute(N, start, kernel):
    X, y, hyp = _data(N, N)
    full = _mk(kernel)
    full.update(X_new=X, y_new=y, hyp=hyp)
    inc = _mk(kernel)
    inc.update(X_new=X[:start], y_new=y[:start], hyp=hyp)
    h0 = inc._post_handle
    for i in range(start, N):
        inc.update(X_new=X[i:i + 1], y_new=y[i:i + 1])
    assert inc._post_handle is h0  # never recomputed: the same device posteriors were extended
    assert np.array_equal(inc.X, full.X) and np.array_equal(inc.y, full.y)
    for a, b in zip(inc.posteriors, full.posteriors):
        assert np.array_equal(a.hyp, b.hyp) and a.sn2_mult == b.sn2_mult and a.L_chol and b.L_chol
        assert a.alpha.shape == (N, 1) and a.L.shape == (N, N) and a.sW.shape == (N, 1)
        assert np.allclose(a.alpha, b.alpha, rtol=1e-7, atol=1e-9 * np.abs(b.alpha).max())
        assert np.allclose(a.sW, b.sW, rtol=1e-12)
        assert np.allclose(a.L, b.L, rtol=1e-8, atol=1e-10)
    xs = np.random.default_rng(1).standard_normal((30, 2))
    m1, v1 = inc.predict(xs, separate_samples=True)
    m2, v2 = full.predict(xs, separate_samples=True)
    assert np.allclose(m1, m2, atol=1e-8) and np.allclose(v1, v2, atol=1e-8)
    n1, g1 = inc.nll_batch(hyp, compute_grad=True)
    n2, g2 = full.nll_batch(hyp, compute_grad=True)
    assert np.array_equal(n1, n2) and np.array_equal(g1, g2)


def test_rank_one_not_applicable_falls_back_to_full_recompute():
    import gpyreg_amd as gpr

    X, y, hyp = _data(30, 3)
    # no constant noise term -> low-noise parametrisation -> full recompute path, same result
    def mk():
        return gpr.GP(2, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                      gpr.noise_functions.GaussianNoise())
    h = hyp[:, [0, 1, 2, 4]]
    a, b = mk(), mk()
    a.update(X_new=X, y_new=y, hyp=h)
    b.update(X_new=X[:29], y_new=y[:29], hyp=h)
    hb = b._post_handle
    b.update(X_new=X[29:], y_new=y[29:])
    assert b._post_handle is not hb  # recomputed
    assert not b.posteriors[0].L_chol
    assert np.allclose(a.posteriors[0].alpha, b.posteriors[0].alpha, rtol=1e-9)
    # two points at once, or a new s2, are never rank-one
    c = _mk()
    c.update(X_new=X[:20], y_new=y[:20], hyp=hyp)
    hc = c._post_handle
    c.update(X_new=X[20:22], y_new=y[20:22])
    assert c._post_handle is not hc and c.posteriors[0].alpha.shape == (22, 1)

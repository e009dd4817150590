"""GPU tests of the drop-in Python API (gpyreg_amd.GP + plugin classes), restating the
properties the reference's own tests pin plus golden parity.

reference tests mirrored:
  test_covariance_functions.py:12-39,54-81,107-126   validation messages
  test_covariance_functions.py:165-215               kernel gradient vs 5-point stencil
  test_isotropic_covariance_functions.py:164-240     iso == ARD with tied length scales
  test_gaussian_process.py:333-346                   NLL gradient vs numeric derivative
  test_gaussian_process.py:254-297                   clean() + update() reproduces posteriors
  test_gaussian_process.py:948-1028                  predict lpd closed form
"""

import numpy as np
import pytest

from conftest import parse_core_name, parse_cov_name

pytestmark = pytest.mark.gpu


def _cov(kernel, degree):
    import gpyreg_amd as gpr

    c, i = gpr.covariance_functions, gpr.isotropic_covariance_functions
    return {
        "se": lambda: c.SquaredExponential(),
        "matern": lambda: c.Matern(degree),
        "rq": lambda: c.RationalQuadraticARD(),
        "se_iso": lambda: i.SquaredExponentialIsotropic(),
        "matern_iso": lambda: i.MaternIsotropic(degree),
    }[kernel]()


def _gp(model, D, dtype="f64"):
    import gpyreg_amd as gpr

    mean = {"zero": gpr.mean_functions.ZeroMean, "const": gpr.mean_functions.ConstantMean,
            "negquad": gpr.mean_functions.NegativeQuadratic}[model["mean"]]()
    p = model["noise"]
    noise = gpr.noise_functions.GaussianNoise(
        constant_add=p[0] == 1, user_provided_add=p[1] >= 1, scale_user_provided=p[1] == 2,
        rectified_linear_output_dependent_add=p[2] == 1)
    return gpr.GP(D, _cov(model["kernel"], model["degree"]), mean, noise, dtype=dtype)


def test_compute_matches_golden(cov_golden):
    g = cov_golden
    for name in g["names"]:
        tag, kernel, degree, N, D, M = parse_cov_name(name)
        cov = _cov(kernel, degree)
        X, Xs, hyp = g[tag + "_X"], g[tag + "_Xs"], g[tag + "_hyp"]
        K, dK = cov.compute(hyp, X, compute_grad=True)
        assert K.shape == (N, N) and dK.shape == (N, N, cov.hyperparameter_count(D))
        assert np.allclose(K, g[tag + "_K"], rtol=1e-12, atol=1e-14), name
        assert np.allclose(dK, g[tag + "_dK"], rtol=1e-11, atol=1e-13, equal_nan=True), name
        assert np.array_equal(np.isnan(dK), np.isnan(g[tag + "_dK"])), name
        assert np.allclose(cov.compute(hyp, X), g[tag + "_K"], rtol=1e-12, atol=1e-14)
        Ks = cov.compute(hyp, X, Xs)
        assert Ks.shape == (N, M) and np.allclose(Ks, g[tag + "_Ks"], rtol=1e-12, atol=1e-14), name
        kd = cov.compute(hyp, Xs, compute_diag=True)
        assert kd.shape == (M, 1) and np.allclose(kd, g[tag + "_kd"], rtol=1e-13), name


def test_compute_validation_messages():
    import gpyreg_amd as gpr

    X = np.ones((10, 3))
    for cov, n in [(gpr.covariance_functions.SquaredExponential(), 4),
                   (gpr.covariance_functions.Matern(3), 4),
                   (gpr.covariance_functions.RationalQuadraticARD(), 5),
                   (gpr.isotropic_covariance_functions.SquaredExponentialIsotropic(), 2),
                   (gpr.isotropic_covariance_functions.MaternIsotropic(5), 2)]:
        with pytest.raises(ValueError) as e:
            cov.compute(np.ones(n + 1), X)
        assert f"Expected {n} covariance function hyperparameters" in e.value.args[0]
        with pytest.raises(ValueError) as e:
            cov.compute(np.ones((n, 1)), X)
        assert "Covariance function output is available only for" in e.value.args[0]
        with pytest.raises(ValueError) as e:
            cov.compute(np.ones(n), X, X_star=X, compute_grad=True)
        assert "X_star should be None when compute_grad is True." in e.value.args[0]
    with pytest.raises(ValueError) as e:
        gpr.covariance_functions.Matern(7)
    assert "Only degrees 1, 3 and 5 are supported for the" in e.value.args[0]


@pytest.mark.parametrize("kernel,degree", [("se", 0), ("matern", 3), ("matern", 5), ("rq", 0),
                                           ("se_iso", 0), ("matern_iso", 5)])
def test_kernel_gradient_five_point_stencil(kernel, degree):
    rng = np.random.default_rng(3)
    N, D = 20, 3
    X = rng.standard_normal((N, D))
    cov = _cov(kernel, degree)
    n = cov.hyperparameter_count(D)
    hyp = rng.standard_normal(n) * 0.3
    _, dK = cov.compute(hyp, X, compute_grad=True)
    h = 1e-5
    for i in range(n):
        e = np.zeros(n)
        e[i] = h
        num = (-cov.compute(hyp + 2 * e, X) + 8 * cov.compute(hyp + e, X)
               - 8 * cov.compute(hyp - e, X) + cov.compute(hyp - 2 * e, X)) / (12 * h)
        assert np.abs(num - dK[:, :, i]).max() < 1e-6


def test_isotropic_equals_ard_with_tied_lengthscales():
    import gpyreg_amd as gpr

    rng = np.random.default_rng(4)
    N, D = 25, 3
    X, Xs = rng.standard_normal((N, D)), rng.standard_normal((7, D))
    for iso, ard in [(gpr.isotropic_covariance_functions.SquaredExponentialIsotropic(),
                      gpr.covariance_functions.SquaredExponential()),
                     (gpr.isotropic_covariance_functions.MaternIsotropic(1), gpr.covariance_functions.Matern(1)),
                     (gpr.isotropic_covariance_functions.MaternIsotropic(5), gpr.covariance_functions.Matern(5))]:
        h_iso = np.array([0.3, -0.2])
        h_ard = np.array([0.3] * D + [-0.2])
        K1, dK1 = iso.compute(h_iso, X, compute_grad=True)
        K2, dK2 = ard.compute(h_ard, X, compute_grad=True)
        assert np.allclose(K1, K2, rtol=1e-12)
        assert np.allclose(dK1[:, :, 0], dK2[:, :, :D].sum(2), rtol=1e-11, atol=1e-13, equal_nan=True)
        assert np.allclose(dK1[:, :, 1], dK2[:, :, D], rtol=1e-12)
        assert np.allclose(iso.compute(h_iso, X, Xs), ard.compute(h_ard, X, Xs), rtol=1e-12)


def test_gp_matches_golden_through_the_api(core_golden):
    g = core_golden
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        if flavour != "plain":
            continue
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        gp = _gp(model, D)
        gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp)
        for s in range(hyp.shape[0]):
            nlZ, dnlZ = gp._GP__compute_nlZ(hyp[s], True, False)
            ref_n, ref_d = g[tag + "_nlZ"][s], g[tag + "_dnlZ"][s]
            assert abs(nlZ - ref_n) <= 1e-8 * max(1.0, abs(ref_n)), name
            ok = ~np.isnan(ref_d)
            assert np.array_equal(np.isnan(dnlZ), ~ok), name
            assert np.abs(dnlZ[ok] - ref_d[ok]).max() <= 1e-8 * np.abs(ref_d[ok]).max(), name
            assert abs(gp.log_likelihood(hyp[s]) + ref_n) <= 1e-8 * max(1.0, abs(ref_n))
            p = gp.posteriors[s]
            assert p.L_chol == bool(g[tag + "_L_chol"][s]) and p.sn2_mult == g[tag + "_sn2_mult"][s]
            assert p.alpha.shape == (N, 1) and p.sW.shape == (N, 1) and p.L.shape == (N, N)
            assert np.allclose(p.alpha[:, 0], g[tag + "_alpha"][s], rtol=1e-7, atol=1e-9 * np.abs(g[tag + "_alpha"][s]).max())
            if p.L_chol:
                assert np.all(np.tril(p.L, -1) == 0)  # upper factor, like SciPy's
        xs, ys = g[tag + "_xs"], g[tag + "_ys"]
        s2s = 0.02 * np.ones((xs.shape[0], 1)) if s2 is not None else None
        mu, v = gp.predict(xs, ys, s2s, separate_samples=True)
        assert np.allclose(mu, g[tag + "_mu_sep"], rtol=1e-8, atol=1e-8), name
        assert np.allclose(v, g[tag + "_s2_sep"], rtol=1e-6, atol=1e-7), name
        mu, v, lpd = gp.predict(xs, ys, s2s, add_noise=True, return_lpd=True)
        assert np.allclose(mu, g[tag + "_mu_avg"], rtol=1e-8, atol=1e-8), name
        assert np.allclose(v, g[tag + "_s2n_avg"], rtol=1e-6, atol=1e-7), name
        assert np.allclose(lpd, g[tag + "_lpd_avg"], rtol=1e-5, atol=1e-6), name
        _, _, lpd = gp.predict(xs, ys, s2s, separate_samples=True, return_lpd=True)
        assert np.allclose(lpd, g[tag + "_lpd_sep"], rtol=1e-5, atol=1e-6), name


def test_nll_gradient_vs_numeric_derivative():
    """reference test_gaussian_process.py:305-346 (numdifftools replaced by a stencil)."""
    import gpyreg_amd as gpr

    rng = np.random.default_rng(11)
    N, D = 20, 2
    X = rng.standard_normal((N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    gp = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    hyp0 = rng.standard_normal(5)
    hyp0[D] *= 0.2
    hyp0[D + 1] *= 0.3
    gp.update(X_new=X, y_new=y, hyp=hyp0[None, :])
    f = lambda h: gp._GP__compute_nlZ(h, False, False)
    _, grad = gp._GP__compute_nlZ(hyp0, True, False)
    h = 1e-4
    for i in range(5):
        e = np.zeros(5)
        e[i] = h
        num = (-f(hyp0 + 2 * e) + 8 * f(hyp0 + e) - 8 * f(hyp0 - e) + f(hyp0 - 2 * e)) / (12 * h)
        assert abs(num - grad[i]) < 1e-6 * max(1.0, abs(grad[i]))


def test_clean_then_update_reproduces_posteriors():
    import gpyreg_amd as gpr

    rng = np.random.default_rng(12)
    N, D = 150, 2
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    hyp = np.array([[0.1, 0.2, 0.0, np.log(0.1), 0.0], [0.3, -0.1, 0.1, np.log(0.2), 0.1]])
    gp = gpr.GP(D, gpr.covariance_functions.Matern(3), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.update(X_new=X, y_new=y, hyp=hyp)
    before = [(p.alpha.copy(), p.sW.copy(), p.L.copy(), p.sn2_mult, p.L_chol) for p in gp.posteriors]
    mu0, s0 = gp.predict(X[:5])
    gp.clean()
    assert all(p.alpha is None and p.L is None and p.sW is None for p in gp.posteriors)
    gp.update(compute_posterior=True)
    for p, (a, w, L, m, lc) in zip(gp.posteriors, before):
        assert np.array_equal(p.alpha, a) and np.array_equal(p.sW, w) and np.array_equal(p.L, L)
        assert p.sn2_mult == m and p.L_chol == lc
    mu1, s1 = gp.predict(X[:5])
    assert np.array_equal(mu0, mu1) and np.array_equal(s0, s1)
    # split update == one update
    gp2 = gpr.GP(D, gpr.covariance_functions.Matern(3), gpr.mean_functions.ConstantMean(),
                 gpr.noise_functions.GaussianNoise(constant_add=True))
    gp2.update(X_new=X[:70], y_new=y[:70], hyp=hyp)
    gp2.update(X_new=X[70:], y_new=y[70:])
    assert np.allclose(gp2.posteriors[1].alpha, before[1][0], rtol=1e-9, atol=1e-11)


def test_predict_lpd_closed_form():
    """reference test_gaussian_process.py:948-1028: lpd == Normal logpdf."""
    import gpyreg_amd as gpr

    rng = np.random.default_rng(13)
    N, D = 60, 2
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    gp = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ZeroMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.update(X_new=X, y_new=y, hyp=np.array([[0.2, 0.1, 0.0, np.log(0.1)]]))
    xs = rng.uniform(-3, 3, (9, D))
    ys = rng.standard_normal((9, 1))
    mu, s2, lpd = gp.predict(xs, ys, add_noise=True, return_lpd=True)
    ref = -0.5 * (ys - mu) ** 2 / s2 - 0.5 * np.log(2 * np.pi * s2)
    assert np.allclose(lpd, ref, rtol=1e-12)
    with pytest.raises(ValueError):
        gp.predict(xs, return_lpd=True)


def test_singular_raises_linalg_error():
    import gpyreg_amd as gpr

    X = np.zeros((40, 1))
    X[:, 0] = np.repeat(np.arange(4.0), 10)  # 10-fold duplicates, no noise term at all
    y = np.ones((40, 1))
    gp = gpr.GP(1, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ZeroMean(),
                gpr.noise_functions.GaussianNoise())
    hyp = np.array([[0.0, 40.0]])  # sigma_f = e^40: jitter eps*1e9 can never rescue it
    with pytest.raises(np.linalg.LinAlgError) as e:
        gp.update(X_new=X, y_new=y, hyp=hyp)
    assert "Singular matrix for L Cholesky decomposition" in str(e.value)


def test_gathered_and_individual_transfers_agree():
    """A call's host arrays travel as ONE gathered transfer (a kernel reading the pinned staging block, which also
    writes the scaled inputs) up to 2 MB; beyond that, segment by segment, through the copy engines, with the
    scaled inputs from their own kernel.  A batch large enough to mix the two paths must give, sample for
    sample, the bits of small batches that fit the gathered path whole (per-point noise and a mean with
    parameters: every optional segment is present)."""
    import gpyreg_amd as gpr

    rng = np.random.default_rng(21)
    N, D, S = 300, 3, 700  # npad = 384: the per-sample vectors alone are 2 x 2.15 MB
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    s2 = 0.01 + 0.02 * rng.random((N, 1))
    gp = gpr.GP(D, gpr.covariance_functions.Matern(5), gpr.mean_functions.NegativeQuadratic(),
                gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True))
    hyp0 = np.concatenate([np.log(1.5) * np.ones(D), [0.0], [np.log(0.1)], [0.0], np.zeros(D), np.zeros(D)])
    hyp = hyp0 + 0.1 * rng.standard_normal((S, hyp0.size))
    gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp[:1], compute_posterior=False)
    big_n, big_g = gp.nll_batch(hyp, compute_grad=True)
    for lo in (0, 350, 693):
        small_n, small_g = gp.nll_batch(hyp[lo:lo + 7], compute_grad=True)
        assert np.array_equal(small_n, big_n[lo:lo + 7])
        assert np.array_equal(small_g, big_g[lo:lo + 7])
    nll_only = gp.nll_batch(hyp, compute_grad=False)[0]
    assert np.array_equal(nll_only[:7], gp.nll_batch(hyp[:7], compute_grad=False)[0])

"""GPU parity of GP.predict_full and GP.quad against reference-generated
vectors (tests/golden/full_cases.npz)."""

import os

import numpy as np
import pytest

from conftest import parse_core_name

pytestmark = pytest.mark.gpu


def test_predict_full_and_quad_match_reference():
    from test_gpu_api import _gp

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "full_cases.npz"), allow_pickle=False)
    for name in g["names"]:
        tag, model, N, D, _ = parse_core_name(str(name) + "|plain")
        X, y, hyp, xs = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"], g[tag + "_xs"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        gp = _gp(model, D)
        gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp)
        s2s = 0.02 * np.ones((xs.shape[0], 1)) if s2 is not None else None
        mu, C = gp.predict_full(xs, None, s2s, add_noise=False)
        assert mu.shape == g[tag + "_pf_mu"].shape and C.shape == g[tag + "_pf_cov"].shape
        assert np.allclose(mu, g[tag + "_pf_mu"], rtol=1e-8, atol=1e-8), name
        scale = np.abs(g[tag + "_pf_cov"]).max()
        assert np.abs(C - g[tag + "_pf_cov"]).max() <= 1e-7 * scale, name
        assert np.array_equal(C, C.transpose(1, 0, 2))
        _, Cn = gp.predict_full(xs, None, s2s, add_noise=True)
        assert np.abs(Cn - g[tag + "_pf_cov_noise"]).max() <= 1e-7 * scale, name
        # diagonal of the full covariance == predictive variance of predict()
        _, v = gp.predict(xs, separate_samples=True)
        assert np.allclose(np.einsum("iis->is", C), v, rtol=1e-6, atol=1e-7 * scale)
        if tag + "_F" in g.files:
            qm, qs = g[tag + "_qm"], g[tag + "_qs"]
            F, Fv = gp.quad(qm, qs, compute_var=True, separate_samples=True)
            assert np.allclose(F, g[tag + "_F"], rtol=1e-8, atol=1e-9), name
            assert np.allclose(Fv, g[tag + "_Fv"], rtol=1e-5, atol=1e-9 * np.abs(g[tag + "_Fv"]).max() + 1e-12), name
            Fa, Fva = gp.quad(qm, qs, compute_var=True)
            assert np.allclose(Fa, g[tag + "_Fa"], rtol=1e-8, atol=1e-9) and Fa.shape == g[tag + "_Fa"].shape
            assert np.allclose(Fva, g[tag + "_Fva"], rtol=1e-5, atol=1e-12)
            assert np.allclose(gp.quad(0.1, 0.5), g[tag + "_F1"], rtol=1e-8, atol=1e-9)
        elif model["kernel"] != "se":
            with pytest.raises(ValueError) as e:
                gp.quad(0.0, 1.0)
            assert "Bayesian quadrature only supports the squared exponential" in str(e.value)


def test_quad_against_numerical_integration():
    """reference test_gaussian_process.py:518-538: quad equals the integral of the posterior
    mean against the Gaussian measure (1-D, trapezoid on a fine grid)."""
    import gpyreg_amd as gpr

    rng = np.random.default_rng(2)
    X = rng.uniform(-3, 3, (60, 1))
    y = np.sin(X) + 0.05 * rng.standard_normal((60, 1))
    gp = gpr.GP(1, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.update(X_new=X, y_new=y, hyp=np.array([[np.log(0.8), 0.0, np.log(0.05), 0.1]]))
    m, sd = 0.4, 0.7
    grid = np.linspace(m - 8 * sd, m + 8 * sd, 4001)[:, None]
    f, _ = gp.predict(grid)
    w = np.exp(-0.5 * ((grid - m) / sd) ** 2) / (sd * np.sqrt(2 * np.pi))
    ref = np.trapezoid((f * w)[:, 0], grid[:, 0])
    F, Fv = gp.quad(np.array([[m]]), np.array([[sd]]), compute_var=True)
    assert abs(F[0, 0] - ref) < 1e-4 and Fv[0, 0] > 0

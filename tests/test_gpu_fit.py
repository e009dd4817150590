"""GPU test of GP.fit: batched design evaluation, start selection,
lock-step multi-start L-BFGS-B, slice sampling, final posteriors -- against a seeded run of
the reference's own fit (tests/golden/fit_cases.npz)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gp(idx, D):
    import gpyreg_amd as gpr

    cov = gpr.covariance_functions.SquaredExponential() if idx == 0 else gpr.covariance_functions.Matern(5)
    return gpr.GP(D, cov, gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True))


@pytest.mark.parametrize("idx", [0, 1])
def test_fit_reproduces_reference_run(idx):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fit_cases.npz"), allow_pickle=False)
    tag = f"f{idx}"
    X, y = g[tag + "_X"], g[tag + "_y"]
    N, D = X.shape
    np.random.seed(1235 + idx)
    X2 = np.random.uniform(low=-3, high=3, size=(N, D))
    y2 = np.reshape(np.sin(np.sum(X2, 1)) + np.random.normal(scale=0.1, size=N), (-1, 1))
    assert np.array_equal(X, X2) and np.array_equal(y, y2)  # RNG state now equals the reference's
    gp = _gp(idx, D)
    gp.set_priors({
        "covariance_log_outputscale": ("student_t", (0, np.log(10), 3)),
        "covariance_log_lengthscale": ("gaussian", (np.log(np.std(X, ddof=1)), np.log(10))),
        "noise_log_scale": ("gaussian", (np.log(1e-3), 1.0)),
        "mean_const": ("smoothbox", (np.min(y), np.max(y), 1.0)),
    })
    opts = {"n_samples": 6, "init_N": 128, "thin": 2, "burn": 12, "opts_N": 3}
    hyp, opt_res, samp = gp.fit(X=X, y=y, options=opts)
    assert np.allclose(gp.lower_bounds, g[tag + "_lb"]) and np.allclose(gp.upper_bounds, g[tag + "_ub"])
    # same design, same start selection, same optimum
    assert abs(opt_res.fun - g[tag + "_opt_fun"]) < 1e-6 * max(1.0, abs(g[tag + "_opt_fun"]))
    assert np.allclose(opt_res.x, g[tag + "_opt_x"], atol=1e-4)
    # the chain consumes the RNG in the same order: same samples up to the optimiser's tolerance
    assert hyp.shape == g[tag + "_hyp"].shape
    assert np.allclose(hyp, g[tag + "_hyp"], atol=2e-3), np.abs(hyp - g[tag + "_hyp"]).max()
    xs = np.random.uniform(-3, 3, size=(15, D))
    assert np.array_equal(xs, g[tag + "_xs"])
    mu, s2 = gp.predict(xs, add_noise=False)
    assert np.allclose(mu, g[tag + "_mu"], atol=5e-3) and np.allclose(s2, g[tag + "_s2"], atol=5e-3)
    assert gp.posteriors.size == 6 and gp.get_hyperparameters(as_array=True).shape == hyp.shape


def test_fit_reproduces_the_reference_example_1():
    """BASELINE config 1 / the reference's own examples/example_1.py: Matern-3 + NegativeQuadratic +
    GaussianNoise(constant_add, user_provided_add) + a Student-t prior on the noise, N = 31, D = 1,
    default fit options (1024-point design, 3 starts, 10 slice samples) -- against the reference's
    seeded run (tests/golden/fit_cases.npz, f2_*)."""
    from scipy.stats import norm

    import gpyreg_amd as gpr

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fit_cases.npz"), allow_pickle=False)
    np.random.seed(1234)
    N, D = 31, 1
    X = -5 + np.random.rand(N, 1) * 10
    s2 = 0.05 * np.exp(0.5 * X)
    y = np.sin(X) + np.sqrt(s2) * norm.ppf(np.random.random_sample(X.shape))
    y[y < 0] = -np.abs(3 * y[y < 0]) ** 2
    assert np.array_equal(X, g["f2_X"]) and np.array_equal(y, g["f2_y"]) and np.array_equal(s2, g["f2_s2"])
    gp = gpr.GP(D=D, covariance=gpr.covariance_functions.Matern(degree=3),
                mean=gpr.mean_functions.NegativeQuadratic(),
                noise=gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True))
    gp.set_priors({"covariance_log_lengthscale": None, "covariance_log_outputscale": None, "mean_const": None,
                   "mean_location": None, "mean_log_scale": None,
                   "noise_log_scale": ("student_t", (np.log(1e-3), 1.0, 7))})
    hyp, opt_res, samp = gp.fit(X=X, y=y, s2=s2, options={"n_samples": 10})
    assert np.allclose(gp.lower_bounds, g["f2_lb"]) and np.allclose(gp.upper_bounds, g["f2_ub"])
    assert abs(opt_res.fun - g["f2_opt_fun"]) < 1e-6 * max(1.0, abs(g["f2_opt_fun"]))
    assert np.allclose(opt_res.x, g["f2_opt_x"], atol=1e-4)
    assert hyp.shape == g["f2_hyp"].shape == (10, 6)
    assert np.allclose(hyp, g["f2_hyp"], atol=5e-3), np.abs(hyp - g["f2_hyp"]).max()
    x_star = np.reshape(np.linspace(-15, 15, 200), (-1, 1))
    fmu, fs2 = gp.predict(x_star, add_noise=False)
    scale = np.abs(g["f2_mu"]).max()
    assert np.abs(fmu - g["f2_mu"]).max() <= 2e-2 * scale and np.abs(fs2 - g["f2_fs2"]).max() <= 2e-2 * np.abs(g["f2_fs2"]).max()


def test_fit_recovers_generating_hyperparameters():
    """reference test_gaussian_process.py:809-849: fit recovers the generating
    hyperparameters (here on the device, N = 500, Matern-5)."""
    import gpyreg_amd as gpr

    np.random.seed(1)
    N, D = 500, 1
    X = np.random.uniform(-5, 5, (N, D))
    hyp_true = np.array([[np.log(1.0), np.log(1.3), np.log(0.1), 0.5]])
    cov = gpr.covariance_functions.Matern(5)
    K = cov.compute(hyp_true[0, :2], X)
    L = np.linalg.cholesky(K + 1e-10 * np.eye(N))
    y = 0.5 + L @ np.random.standard_normal((N, 1)) + 0.1 * np.random.standard_normal((N, 1))
    gp = gpr.GP(D, cov, gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True))
    hyp, opt_res, _ = gp.fit(X=X, y=y, options={"n_samples": 0, "init_N": 256})
    assert hyp.shape == (1, 4)
    assert np.abs(hyp[0, :3] - hyp_true[0, :3]).max() < 0.5
    ll_fit, ll_true = gp.log_likelihood(hyp[0]), gp.log_likelihood(hyp_true[0])
    assert ll_fit >= ll_true - 1e-6 and ll_fit - ll_true < 20


def test_example_script_runs_end_to_end():
    """BASELINE config 1 (plumbing): the example runs through fit/predict/update."""
    import importlib.util

    path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "fit_predict_2d.py")
    spec = importlib.util.spec_from_file_location("fit_predict_2d", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gp, fmu, fs2, fmu2, fs22 = mod.main(verbose=False)
    assert fmu.shape == (400, 1) and fs2.shape == (400, 1) and np.all(fs2 >= 0)
    assert gp.X.shape == (40, 2) and gp.posteriors.size == 10
    assert np.sqrt(fs22).mean() < np.sqrt(fs2).mean()  # more data, less uncertainty

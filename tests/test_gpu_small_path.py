"""Problems that are ONE 128 x 128 leaf (N <= 128) take a pipeline of their own (gpcore.hip: Pipe::small_section --
upload + covariance build in one launch, factorization + both triangular products in one launch; the regime of
f_min_fill.py:174-176 and slice_sample.py:442 on small training sets).  It must agree with the general pipeline (the
same leaf arithmetic; the products add their terms in another order: rounding only), with the CPU oracle at 1e-8, and a
row of a batch must carry the bits of its single evaluation."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gpyreg_amd import _lib

    return _lib.context(0)


def _problem(N, D, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
    return rng, X, y


def _models():
    import gpyreg_amd as gpr

    cf, iso, mf = gpr.covariance_functions, gpr.isotropic_covariance_functions, gpr.mean_functions
    return [("se", lambda: cf.SquaredExponential(), dict(kernel="se", degree=0)),
            ("matern5", lambda: cf.Matern(5), dict(kernel="matern", degree=5)),
            ("matern3", lambda: cf.Matern(3), dict(kernel="matern", degree=3)),
            ("rq", lambda: cf.RationalQuadraticARD(), dict(kernel="rq", degree=0)),
            ("se_iso", lambda: iso.SquaredExponentialIsotropic(), dict(kernel="se_iso", degree=0))], mf


@pytest.mark.parametrize("N", [5, 33, 100, 128])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_small_path_agrees_with_the_general_pipeline_and_the_oracle(ctx, N, dtype):
    import gpyreg_amd as gpr
    from oracle import gp_oracle as orc  # checker only

    models, mf = _models()
    tol_paths = 1e-12 if dtype == "f64" else 2e-5
    for name, mk, om in models:
        D = 3
        rng, X, y = _problem(N, D, 11 + N)
        cov = mk()
        gp = gpr.GP(D, cov, mf.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True), dtype=dtype)
        cov_N = cov.hyperparameter_count(D)
        S = 5
        hyp = np.concatenate([np.log(1.5) * np.ones(min(D, cov_N - 1)), np.zeros(cov_N - min(D, cov_N - 1)),
                              [np.log(0.2), 0.1]]) + 0.1 * rng.standard_normal((S, cov_N + 2))
        gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
        assert ctx.get_option("small_path") == 1
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
        n0, _ = gp.nll_batch(hyp, compute_grad=False)
        assert np.array_equal(n0, nlz)  # with and without gradient: the same value
        for s in (0, S - 1):  # a row of a batch == its single evaluation, bit for bit
            n1, d1 = gp.nll_batch(hyp[s:s + 1], compute_grad=True)
            m1, _ = gp.nll_batch(hyp[s:s + 1], compute_grad=False)
            assert n1[0] == nlz[s] and m1[0] == nlz[s] and np.array_equal(d1[0], dnlz[s]), (name, s)
        ctx.set_option("small_path", 0)
        try:
            gn, gd = gp.nll_batch(hyp, compute_grad=True)
        finally:
            ctx.set_option("small_path", 1)
        assert np.abs(gn - nlz).max() <= tol_paths * np.abs(gn).max(), (name, gn, nlz)
        assert np.abs(gd - dnlz).max() <= tol_paths * 100 * np.abs(gd).max(), (name, np.abs(gd - dnlz).max())
        if dtype == "f64":
            model = dict(om, mean="const", noise=(1, 0, 0))
            for s in range(S):
                rn, rd = orc.core(model, hyp[s], X, y, None, 1, 1)
                assert abs(nlz[s] - rn) <= 1e-8 * max(1.0, abs(rn)), (name, s, nlz[s], rn)
                assert np.abs(dnlz[s] - rd).max() <= 1e-8 * max(1.0, np.abs(rd).max()), (name, s)


def test_small_path_with_general_mean_and_per_point_noise(ctx):
    """NegativeQuadratic mean (mean gradient products) and user-provided + output-dependent noise (vector noise and its
    gradient products) through the small pipeline, against the oracle."""
    import gpyreg_amd as gpr
    from oracle import gp_oracle as orc  # checker only

    N, D, S = 90, 2, 3
    rng, X, y = _problem(N, D, 5)
    s2 = 0.01 + 0.02 * rng.uniform(size=(N, 1))
    noise = gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True, scale_user_provided=True,
                                              rectified_linear_output_dependent_add=True)
    gp = gpr.GP(D, gpr.covariance_functions.Matern(5), gpr.mean_functions.NegativeQuadratic(), noise)
    hyp = np.concatenate([np.log(1.2) * np.ones(D), [0.0], [np.log(0.1), 0.0, float(np.max(y)) - 0.3, np.log(0.05)],
                          [0.2], np.zeros(D), np.log(3.0) * np.ones(D)]) + 0.05 * rng.standard_normal((S, 3 + 4 + 1 + 2 * D))
    gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    model = dict(kernel="matern", degree=5, mean="negquad", noise=(1, 2, 1))
    for s in range(S):
        rn, rd = orc.core(model, hyp[s], X, y, s2, 1, 1)
        assert abs(nlz[s] - rn) <= 1e-8 * max(1.0, abs(rn))
        assert np.abs(dnlz[s] - rd).max() <= 1e-8 * max(1.0, np.abs(rd).max())


def test_small_path_jitter_retry_and_failure(ctx):
    """A numerically singular small system goes through the small pipeline first, fails there, and is retried by the
    general stable-mode pipeline: same multiplier and value as with the small pipeline switched off."""
    import gpyreg_amd as gpr

    N, D = 60, 2
    rng, X, y = _problem(N, D, 3)
    X[30:] = X[:30]  # duplicated inputs: singular without noise
    gp = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    hyp = np.array([[np.log(2.0), np.log(2.0), 0.0, np.log(1e-9), 0.0]])
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    a, _ = gp.nll_batch(hyp, compute_grad=False)
    ctx.set_option("small_path", 0)
    try:
        b, _ = gp.nll_batch(hyp, compute_grad=False)
    finally:
        ctx.set_option("small_path", 1)
    assert np.isfinite(a).all() and abs(a[0] - b[0]) <= 1e-6 * abs(b[0])


def test_a_leaf_time_out_in_the_small_pipeline_is_an_error(ctx):
    import gpyreg_amd as gpr

    N, D = 40, 2
    rng, X, y = _problem(N, D, 4)
    gp = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    hyp = np.array([[np.log(2.0), np.log(2.0), 0.0, np.log(0.1), 0.0]])
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    ctx.set_option("leaf_fault", 1)
    try:
        with pytest.raises(RuntimeError, match="timed out"):
            gp.nll_batch(hyp, compute_grad=False)
    finally:
        ctx.set_option("leaf_fault", 0)
    n, _ = gp.nll_batch(hyp, compute_grad=False)
    assert np.isfinite(n).all()


def test_fused_predict_product_in_fp32_against_fp64():
    """The predict product whose epilogue forms the variance sums (gemm.h: EPI = 1) with the fp32 two-level accumulators:
    N = 1500, M = 700 (12 x 6 tiles: the fused form), fp32 mode against fp64 mode at north_star's 1e-3, and fp64 against
    the CPU oracle."""
    import gpyreg_amd as gpr
    from oracle import gp_oracle as orc  # checker only

    N, D, S, M = 1500, 4, 3, 700
    rng, X, y = _problem(N, D, 77)
    hyp = np.concatenate([np.log(1.5) * np.ones(D), [0.0, np.log(0.2), 0.1]]) + 0.1 * rng.standard_normal((S, D + 3))
    xs = rng.uniform(-3, 3, (M, D))
    res = {}
    for dtype in ("f64", "f32"):
        gp = gpr.GP(D, gpr.covariance_functions.Matern(5), gpr.mean_functions.ConstantMean(),
                    gpr.noise_functions.GaussianNoise(constant_add=True), dtype=dtype)
        gp.update(X_new=X, y_new=y, hyp=hyp)
        res[dtype] = gp.predict(xs, separate_samples=True)
        one = gpr.GP(D, gpr.covariance_functions.Matern(5), gpr.mean_functions.ConstantMean(),
                     gpr.noise_functions.GaussianNoise(constant_add=True), dtype=dtype)
        one.update(X_new=X, y_new=y, hyp=hyp[1:2])
        m1, v1 = one.predict(xs, separate_samples=True)
        assert np.array_equal(m1[:, 0], res[dtype][0][:, 1]) and np.array_equal(v1[:, 0], res[dtype][1][:, 1])
    model = dict(kernel="matern", degree=5, mean="const", noise=(1, 0, 0))
    posts = orc.posteriors(model, hyp, X, y, None)
    rmu, rs2 = orc.predict(model, posts, X, y, xs, separate_samples=True)
    assert np.abs(res["f64"][0] - rmu).max() < 1e-8 and np.abs(res["f64"][1] - rs2).max() < 1e-8 * max(1.0, rs2.max())
    assert np.abs(res["f32"][0] - rmu).max() < 1e-3 * max(1.0, np.abs(rmu).max())
    assert np.abs(res["f32"][1] - rs2).max() < 1e-3 * max(1.0, rs2.max())


@pytest.mark.parametrize("N", [100, 300])
def test_many_input_dimensions_through_the_new_kernels(ctx, N):
    """D = 70: more dimensions than one LDS stage holds (32: the distance sweeps restage), more than the one-leaf
    pipeline keeps scaling factors for in LDS (64: read from the staged host copy), and an odd count for the packed
    fp32 sweeps -- NLL and gradient against the oracle in fp64, fp32 against fp64 at 1e-3, predict likewise."""
    import gpyreg_amd as gpr
    from oracle import gp_oracle as orc  # checker only

    D, S = 71, 3
    rng = np.random.default_rng(N + 7)
    X = rng.uniform(-1, 1, (N, D))
    y = np.sin(X[:, :3].sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    hyp = np.concatenate([np.log(6.0) * np.ones(D), [0.0, np.log(0.2), 0.1]]) + 0.05 * rng.standard_normal((S, D + 3))
    xs = rng.uniform(-1, 1, (9, D))
    res = {}
    for dtype in ("f64", "f32"):
        gp = gpr.GP(D, gpr.covariance_functions.Matern(3), gpr.mean_functions.ConstantMean(),
                    gpr.noise_functions.GaussianNoise(constant_add=True), dtype=dtype)
        gp.update(X_new=X, y_new=y, hyp=hyp)
        res[dtype] = gp.nll_batch(hyp, compute_grad=True) + gp.predict(xs, separate_samples=True)
    model = dict(kernel="matern", degree=3, mean="const", noise=(1, 0, 0))
    posts = orc.posteriors(model, hyp, X, y, None)
    rmu, rs2 = orc.predict(model, posts, X, y, xs, separate_samples=True)
    nlz, dnlz, mu, s2 = res["f64"]
    for s in range(S):
        rn, rd = orc.core(model, hyp[s], X, y, None, 1, 1)
        assert abs(nlz[s] - rn) <= 1e-8 * max(1.0, abs(rn))
        assert np.abs(dnlz[s] - rd).max() <= 1e-8 * max(1.0, np.abs(rd).max())
    assert np.abs(mu - rmu).max() < 1e-8 and np.abs(s2 - rs2).max() < 1e-8
    n32, d32, mu32, s232 = res["f32"]
    assert np.abs(n32 - nlz).max() <= 1e-3 * np.abs(nlz).max()
    assert np.abs(d32 - dnlz).max() <= 1e-3 * np.abs(dnlz).max()
    assert np.abs(mu32 - mu).max() <= 1e-3 and np.abs(s232 - s2).max() <= 1e-3 * max(1.0, s2.max())

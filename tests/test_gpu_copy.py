"""copy.deepcopy and pickle of a GP (the reference's GP is plain Python and its users -- PyVBMC -- copy and store it):
the host state travels, the device posteriors are rebuilt on first use; copies are independent of the original."""

import copy
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fitted(S=3, N=70, seed=5):
    import gpyreg_amd as gpr

    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, (N, 2))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    hyp = np.array([0.2, 0.3, 0.1, np.log(0.1), 0.05]) + 0.1 * rng.standard_normal((S, 5))
    gp = gpr.GP(2, gpr.covariance_functions.Matern(5), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.set_priors({"covariance_log_lengthscale": None, "covariance_log_outputscale": None, "mean_const": None,
                   "noise_log_scale": ("gaussian", (np.log(1e-2), 1.0))})
    gp.update(X_new=X, y_new=y, hyp=hyp)
    return gp, rng


@pytest.mark.parametrize("how", ["deepcopy", "pickle"])
def test_a_copy_predicts_what_the_original_predicts(how):
    gp, rng = _fitted()
    xs = rng.uniform(-3, 3, (11, 2))
    mu, s2 = gp.predict(xs, separate_samples=True)
    a0 = gp.posteriors[0].alpha  # materialised on the host before the copy: travels as it is
    cp = copy.deepcopy(gp) if how == "deepcopy" else pickle.loads(pickle.dumps(gp))
    assert cp._post_handle is None and cp._rebuild and cp.posteriors.size == 3
    assert np.array_equal(cp.posteriors[0].alpha, a0) and cp._post_handle is None  # no device work for that
    assert np.array_equal(cp.X, gp.X) and cp.X is not gp.X
    assert np.array_equal(cp.get_hyperparameters(as_array=True), gp.get_hyperparameters(as_array=True))
    p1 = cp.posteriors[1]
    m2, v2 = cp.predict(xs, separate_samples=True)  # rebuilds the device posteriors
    assert cp._post_handle is not None and cp._post_handle is not gp._post_handle and not cp._rebuild
    assert np.array_equal(m2, mu) and np.array_equal(v2, s2)  # the same full factorizations: the same bits
    assert cp.posteriors[1] is p1 and np.array_equal(p1.alpha, gp.posteriors[1].alpha)  # records keep their identity
    assert np.array_equal(np.asarray(cp.posteriors[2].L), np.asarray(gp.posteriors[2].L))
    assert cp.log_posterior(gp.posteriors[0].hyp) == gp.log_posterior(gp.posteriors[0].hyp)  # priors travelled
    # independent objects: the copy grows, the original does not notice
    cp.update(X_new=xs[:1], y_new=np.zeros((1, 1)))
    assert cp.X.shape[0] == gp.X.shape[0] + 1
    m3, v3 = gp.predict(xs, separate_samples=True)
    assert np.array_equal(m3, mu) and np.array_equal(v3, s2)


def test_copies_of_other_states():
    import gpyreg_amd as gpr

    gp, rng = _fitted(S=2)
    xs = rng.uniform(-3, 3, (5, 2))
    # a set that grew by rank-one appends is rebuilt by full factorizations: equal to rounding
    for k in range(3):
        gp.update(X_new=xs[k:k + 1], y_new=np.array([[0.1 * k]]))
    mu, s2 = gp.predict(xs, separate_samples=True)
    cp = copy.deepcopy(gp)
    m2, v2 = cp.predict(xs, separate_samples=True)
    assert np.allclose(m2, mu, rtol=1e-9, atol=1e-11) and np.allclose(v2, s2, rtol=1e-8, atol=1e-11)
    # the first use of a copy may be a one-point update (the resident posteriors are rebuilt first, then extended)
    cp2 = copy.deepcopy(gp)
    cp2.update(X_new=xs[3:4], y_new=np.array([[0.3]]))
    gp.update(X_new=xs[3:4], y_new=np.array([[0.3]]))
    assert np.allclose(cp2.predict(xs)[0], gp.predict(xs)[0], rtol=1e-9, atol=1e-11)
    # a cleaned GP stays cleaned
    gp.clean()
    cl = pickle.loads(pickle.dumps(gp))
    assert not cl._rebuild and cl.posteriors[0].alpha is None
    with pytest.raises(ValueError, match="cleaned"):
        cl.predict(xs)
    # a copy of a copy that was never used still rebuilds
    gp3, _ = _fitted(S=2, seed=9)
    ref = gp3.predict(xs, separate_samples=True)
    cc = copy.deepcopy(copy.deepcopy(gp3))
    out = cc.predict(xs, separate_samples=True)
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
    # one record by itself: the reference's plain record, fields on the host
    rec = copy.deepcopy(gp3.posteriors[1])
    assert rec._handle is None and np.array_equal(rec.alpha, gp3.posteriors[1].alpha)
    assert np.array_equal(np.asarray(rec.L), np.asarray(gp3.posteriors[1].L))
    # a GP without data
    g0 = gpr.GP(2, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ZeroMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    g0.update(hyp=np.array([[0.1, 0.2, 0.0, np.log(0.1)]]))
    c0 = copy.deepcopy(g0)
    np.random.seed(3)
    f0 = g0.random_function(xs)
    np.random.seed(3)
    assert np.array_equal(c0.random_function(xs), f0)

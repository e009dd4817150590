"""GPU tests of the rank-one posterior append: against the reference's own
rank-one path (tests/golden/rank1_cases.npz, produced by gpyreg's GP.update with one new point,
gaussian_process.py:750-844, high- and low-noise parametrisation), against the oracle's full
recompute for a posterior declared unstable (the reference's per-posterior fallback, :789-798),
and the reference's property test "rank-1 updates == full recompute"
(test_gaussian_process.py:387-411)."""

import os

import numpy as np
import pytest

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "rank1_cases.npz"), allow_pickle=False)


def _parse(name):
    tag, kname, mname, npar, N, D, flav = str(name).split("|")
    degree, kernel = 0, kname
    if kname.startswith("matern"):
        kernel, degree = "matern", int(kname[6:])
    return tag, dict(kernel=kernel, degree=degree, mean=mname, noise=tuple(int(c) for c in npar)), int(N), int(D), flav


def _check_L(p, g, tag, s, tol):
    L = np.asarray(p.L)
    for key, mine in (("_Ldiag", np.diag(L)), ("_Llast_col", L[:, -1]), ("_Llast_row", L[-1, :])):
        ref = g[tag + key][s]
        assert np.abs(mine - ref).max() <= tol * max(np.abs(ref).max(), 1e-300), (tag, s, key)
    assert abs(np.linalg.norm(L) - g[tag + "_Lfro"][s]) <= tol * g[tag + "_Lfro"][s], (tag, s)


def test_rank_one_appends_match_the_reference_rank_one_path():
    """Three consecutive one-point updates; predictions after each and alpha / sW / L after the
    last against the reference's rank-one results, 1e-8 relative.  Includes the storage growth
    across a 128-tile boundary (N = 126, 127, 128 -> +3) and the low-noise branch (:819-827)."""
    from test_gpu_api import _gp as mk

    g = _golden()
    for name in g["names"]:
        tag, model, N, D, flav = _parse(name)
        X, y, hyp, xs = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"], g[tag + "_xs"]
        Xn, yn = g[tag + "_Xn"], g[tag + "_yn"]
        gp = mk(model, D)
        gp.update(X_new=X, y_new=y, hyp=hyp)
        h0 = gp._post_handle
        assert all(bool(p.L_chol) == (flav == "high") for p in gp.posteriors)
        for k in range(3):
            gp.update(X_new=Xn[k:k + 1], y_new=yn[k:k + 1])
            assert gp._post_handle is h0, (name, "the resident posteriors must be extended, not rebuilt")
            mu, s2 = gp.predict(xs, separate_samples=True)
            rm, rs = g[tag + f"_mu{k}"], g[tag + f"_s2{k}"]
            assert np.abs(mu - rm).max() <= 1e-8 * max(1.0, np.abs(rm).max()), (name, k, "mu")
            assert np.abs(s2 - rs).max() <= 1e-8 * max(1.0, np.abs(rs).max()), (name, k, "s2")
        assert gp.X.shape[0] == N + 3
        for s, p in enumerate(gp.posteriors):
            ra = g[tag + "_alpha"][s]
            assert p.alpha.shape == (N + 3, 1) and p.sW.shape == (N + 3, 1) and p.L.shape == (N + 3, N + 3)
            assert np.abs(p.alpha[:, 0] - ra).max() <= 1e-8 * np.abs(ra).max(), (name, s, "alpha")
            assert np.allclose(p.sW[:, 0], g[tag + "_sW"][s], rtol=1e-12), (name, s, "sW")
            assert bool(p.L_chol) == bool(g[tag + "_L_chol"][s])
            _check_L(p, g, tag, s, 1e-8)


def test_unstable_posterior_alone_is_recomputed():
    """The reference recomputes only the posterior whose append is unstable (full_updates,
    :789-798, :866-869).  Sample 1 of 3 is declared unstable through the test hook: it must equal
    the oracle's full recompute on the extended data, the others the reference's rank-one values."""
    from gpyreg_amd import _lib
    from test_gpu_api import _gp as mk

    g = _golden()
    ctx = _lib.context(0)
    for name in g["names"]:
        tag, model, N, D, flav = _parse(name)
        if N not in (33, 128):
            continue
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        Xn, yn = g[tag + "_Xn"], g[tag + "_yn"]
        gp = mk(model, D)
        gp.update(X_new=X, y_new=y, hyp=hyp)
        ctx.set_option("append_fail_mask", 0b010)
        try:
            gp.update(X_new=Xn[:1], y_new=yn[:1])
        finally:
            ctx.set_option("append_fail_mask", 0)
        X1, y1 = np.concatenate([X, Xn[:1]]), np.concatenate([y, yn[:1]])
        full = orc.posteriors(model, hyp, X1, y1, None)
        ref_mu, ref_s2 = orc.predict(model, full, X1, y1, g[tag + "_xs"], separate_samples=True)
        mu, s2 = gp.predict(g[tag + "_xs"], separate_samples=True)
        # sample 1: the full recompute; samples 0, 2: rank-one == full recompute to rounding as well
        assert np.abs(mu - ref_mu).max() <= 1e-7 * max(1.0, np.abs(ref_mu).max()), name
        assert np.abs(s2 - ref_s2).max() <= 1e-7 * max(1.0, np.abs(ref_s2).max()), name
        p1 = gp.posteriors[1]
        assert np.abs(p1.alpha - full[1].alpha).max() <= 1e-8 * np.abs(full[1].alpha).max(), name
        assert np.abs(np.asarray(p1.L) - full[1].L).max() <= 1e-8 * np.abs(full[1].L).max(), name
        assert p1.sn2_mult == full[1].sn2_mult and bool(p1.L_chol) == bool(full[1].L_chol)
        assert np.abs(mu[:, [0, 2]] - g[tag + "_mu0"][:, [0, 2]]).max() <= 1e-8 * max(1.0, np.abs(ref_mu).max()), name


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
    # output-dependent (per-point) noise: the append formulas do not apply -> full recompute
    def mk():
        return gpr.GP(2, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                      gpr.noise_functions.GaussianNoise(constant_add=True, rectified_linear_output_dependent_add=True))
    h = np.concatenate([hyp[:, :4], np.array([[0.0, np.log(0.05)], [0.1, np.log(0.04)]]), hyp[:, 4:]], axis=1)
    a, b = mk(), mk()
    a.update(X_new=X, y_new=y, hyp=h)
    b.update(X_new=X[:29], y_new=y[:29], hyp=h)
    hb = b._post_handle
    b.update(X_new=X[29:], y_new=y[29:])
    assert b._post_handle is not hb  # recomputed
    assert np.allclose(a.posteriors[0].alpha, b.posteriors[0].alpha, rtol=1e-9)
    # two points at once, or a new s2, are never rank-one
    c = _mk()
    c.update(X_new=X[:20], y_new=y[:20], hyp=hyp)
    hc = c._post_handle
    c.update(X_new=X[20:22], y_new=y[20:22])
    assert c._post_handle is not hc and c.posteriors[0].alpha.shape == (22, 1)


def test_rank_one_appends_with_a_user_defined_kernel_match_the_reference():
    """The reference's rank-one path calls ``self.covariance.compute`` whatever the object is
    (gaussian_process.py:771-772).  A Python SE kernel (no device code: K and the cross covariances come from its own
    compute(), gpc_posterior_batch_K / gpc_post_append_K) must reproduce the reference's rank-one results of the
    built-in SE fixtures: predictions after each of three appended points and alpha / sW / L after the last, 1e-8;
    and a posterior declared unstable is recomputed alone from the object's own K (gpc_post_recompute_K)."""
    import gpyreg_amd as gpr
    from gpyreg_amd import _lib
    from test_gpu_user_kernel import PySquaredExponential

    g = _golden()
    ran = 0
    for name in g["names"]:
        tag, model, N, D, flav = _parse(name)
        if model["kernel"] != "se" or model["noise"] != (1, 0, 0):
            continue
        ran += 1
        Mean = {"const": gpr.mean_functions.ConstantMean, "negquad": gpr.mean_functions.NegativeQuadratic,
                "zero": gpr.mean_functions.ZeroMean}[model["mean"]]
        X, y, hyp, xs = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"], g[tag + "_xs"]
        Xn, yn = g[tag + "_Xn"], g[tag + "_yn"]
        gp = gpr.GP(D, PySquaredExponential(), Mean(), gpr.noise_functions.GaussianNoise(constant_add=True))
        assert not gp._builtin
        gp.update(X_new=X, y_new=y, hyp=hyp)
        h0 = gp._post_handle
        for k in range(3):
            gp.update(X_new=Xn[k:k + 1], y_new=yn[k:k + 1])
            assert gp._post_handle is h0, (name, "the resident posteriors must be extended, not rebuilt")
            mu, s2 = gp.predict(xs, separate_samples=True)
            rm, rs = g[tag + f"_mu{k}"], g[tag + f"_s2{k}"]
            assert np.abs(mu - rm).max() <= 1e-8 * max(1.0, np.abs(rm).max()), (name, k, "mu")
            assert np.abs(s2 - rs).max() <= 1e-8 * max(1.0, np.abs(rs).max()), (name, k, "s2")
        for s, p in enumerate(gp.posteriors):
            ra = g[tag + "_alpha"][s]
            assert np.abs(p.alpha[:, 0] - ra).max() <= 1e-8 * np.abs(ra).max(), (name, s, "alpha")
            assert np.allclose(p.sW[:, 0], g[tag + "_sW"][s], rtol=1e-12), (name, s, "sW")
            _check_L(p, g, tag, s, 1e-8)
        # the per-posterior fallback with the object's own K
        gp2 = gpr.GP(D, PySquaredExponential(), Mean(), gpr.noise_functions.GaussianNoise(constant_add=True))
        gp2.update(X_new=X, y_new=y, hyp=hyp)
        ctx = _lib.context(0)
        ctx.set_option("append_fail_mask", 0b010)
        try:
            gp2.update(X_new=Xn[:1], y_new=yn[:1])
        finally:
            ctx.set_option("append_fail_mask", 0)
        X1, y1 = np.concatenate([X, Xn[:1]]), np.concatenate([y, yn[:1]])
        full = orc.posteriors(model, hyp, X1, y1, None)
        p1 = gp2.posteriors[1]
        assert np.abs(p1.alpha - full[1].alpha).max() <= 1e-8 * np.abs(full[1].alpha).max(), name
        assert np.abs(np.asarray(p1.L) - full[1].L).max() <= 1e-8 * np.abs(full[1].L).max(), name
        mu, s2 = gp2.predict(xs, separate_samples=True)
        assert np.abs(mu - g[tag + "_mu0"]).max() <= 1e-7 * max(1.0, np.abs(g[tag + "_mu0"]).max()), name
    assert ran >= 2

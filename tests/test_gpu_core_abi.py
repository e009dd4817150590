"""GPU parity of the core path through the C ABI against the reference's golden
vectors (tests/golden/core_cases.npz): gpc_nll_batch, gpc_posterior_batch,
gpc_post_fetch, gpc_predict.  Mean/noise plugin values (O(N*D) host inputs of the
ABI) come from the oracle here so that this file tests the device code alone.

Tolerance (north_star): 1e-8 relative for fp64 NLL and gradient, 1e-3 for fp32.
Gradient components are compared relative to max(|ref_i|, ||ref||_inf) because a
component may legitimately be ~0.  Ill-conditioned fixtures (no-noise / jitter
flavours, cond(A) = 1e7 ... 1e17) are compared against an extended-precision
evaluation at max(1e-8, 8 cond eps) and must reproduce the branch flags; no fixture
passes without a value check (see test_nll_and_grad_match_golden).
"""

import numpy as np
import pytest

from conftest import parse_core_name
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

KID = {"se": 0, "matern": 1, "rq": 2, "se_iso": 3, "matern_iso": 4}


@pytest.fixture(scope="module")
def ctx():
    from gpyreg_amd import _lib

    return _lib.context(0)


def plugin_inputs(model, hyp, X, y, s2, grad):
    """m, sn2, dm, dsn2 stacked over samples, exactly what the ABI takes."""
    S = hyp.shape[0]
    N, D = X.shape
    cov_N = orc.cov_count(model["kernel"], D)
    noise_N = orc.noise_count(model["noise"])
    mean_N = orc.mean_count(model["mean"], D)
    ms, sn2s, dms, dsn2s = [], [], [], []
    vec = None
    for s in range(S):
        hn = hyp[s, cov_N:cov_N + noise_N]
        hm = hyp[s, cov_N + noise_N:]
        if grad:
            sn2, dsn2 = orc.noise(model["noise"], hn, X, y, s2, compute_grad=True)
            m, dm = orc.mean(model["mean"], hm, X, compute_grad=True)
            dms.append(np.zeros((N, 0)) if mean_N == 0 else np.asarray(dm))
            dsn2s.append(dsn2)
        else:
            sn2 = orc.noise(model["noise"], hn, X, y, s2)
            m = orc.mean(model["mean"], hm, X)
        vec = not np.isscalar(sn2)
        sn2s.append(np.ravel(sn2) if vec else np.array([sn2]))
        ms.append(m)
    out = dict(m=np.stack(ms), sn2=np.stack(sn2s), vec=vec, cov_N=cov_N)
    if grad:
        out["dm"] = np.stack(dms) if mean_N else None
        out["dsn2"] = np.stack(dsn2s) if noise_N else None
    return out


def rel_vec(a, b):
    scale = np.maximum(np.abs(b), np.nanmax(np.abs(b)) if np.isfinite(b).any() else 1.0)
    return np.abs(a - b) / scale


EPS = np.finfo(float).eps


def _errs(nlz, dnlz, ref_n, ref_d, gscale=None):
    """(relative nlZ error, relative gradient error) of one sample; NaN gradient entries (Matern-1)
    must coincide and are left out.  ``gscale`` (ill-conditioned fixtures only): per component,
    cond * eps times the absolute sum of the terms the component is a signed sum of
    (orc.core_extended): on singular systems those terms are ~1e19 and cancel to ~10, so no
    arithmetic resolves the component itself; the gradient error is then measured against
    max(|ref_i|, ||ref||_inf, gscale_i) -- tighter than the first-order bound by a factor cond * eps."""
    assert np.array_equal(np.isnan(dnlz), np.isnan(ref_d))
    ok = ~np.isnan(ref_d)
    e_n = abs(nlz - ref_n) / max(abs(ref_n), 1.0)
    if not ok.any():
        return e_n, 0.0
    if gscale is None:
        return e_n, rel_vec(dnlz[ok], ref_d[ok]).max()
    scale = np.maximum(np.maximum(np.abs(ref_d[ok]), np.abs(ref_d[ok]).max()), gscale[ok])
    return e_n, (np.abs(dnlz[ok] - ref_d[ok]) / scale).max()


def test_nll_and_grad_match_golden(ctx, core_golden):
    """Every fixture gets a VALUE check; none passes on control flow alone.

    * well-conditioned ("plain") fixtures: the reference's golden nlZ / dnlZ at 1e-8 relative
      (north_star), same jitter multiplier, same branch flag;
    * ill-conditioned fixtures (low noise, jitter escalation; cond(A) = 1e7 ... 1e17): an fp64
      result is only defined to about cond(A) * eps.  The yardstick is the oracle's
      extended-precision evaluation at the multiplier the DEVICE settled on
      (``orc.core_extended``): the device must be within 8 * cond * eps of it (floor 1e-8) --
      LAPACK itself, i.e. the reference's golden value, is within 0.6 * cond * eps on these
      fixtures (tests/test_oracle_golden.py).  Where the device's first successful jitter level
      equals LAPACK's the golden value is compared as well, at the same bar; where it differs
      (rounding dependent on singular matrices) the pinned oracle is evaluated with
      ``force_mult`` = the device's level and compared when LAPACK succeeds there.
    """
    from gpyreg_amd import _lib

    g = core_golden
    report, worst, n_diff_mult, above, below = [], 0.0, 0, [], []
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        pin = plugin_inputs(model, hyp, X, y, s2, True)
        ctx.set_data(X, y)
        nlz, dnlz, mult, lchol, info = ctx.nll_batch(
            KID[model["kernel"]], model["degree"], _lib.F64, hyp[:, :pin["cov_N"]], pin["m"], pin["sn2"],
            pin["vec"], True, pin["dm"], pin["dsn2"])
        nlz0, *_ = ctx.nll_batch(KID[model["kernel"]], model["degree"], _lib.F64, hyp[:, :pin["cov_N"]],
                                 pin["m"], pin["sn2"], pin["vec"], False)
        ref_n, ref_d = g[tag + "_nlZ"], g[tag + "_dnlZ"]
        assert (info == 0).all(), name
        assert np.array_equal(lchol, g[tag + "_L_chol"]), name
        gm = g[tag + "_sn2_mult"]
        plain = flavour == "plain"
        if plain:
            assert np.array_equal(mult, gm), (name, mult, gm)
        for s in range(hyp.shape[0]):
            same = mult[s] == gm[s]
            n_diff_mult += not same
            (above if mult[s] > gm[s] else below if mult[s] < gm[s] else []).append("%s s=%d: %g vs %g" % (tag, s, mult[s], gm[s]))
            checked = []
            assert abs(nlz0[s] - nlz[s]) <= 1e-12 * max(1.0, abs(nlz[s])), name  # NLL-only path == NLL of NLL+grad
            if plain:
                e_n, e_d = _errs(nlz[s], dnlz[s], ref_n[s], ref_d[s])
                e_0 = abs(nlz0[s] - g[tag + "_nlZ_only"][s]) / max(abs(ref_n[s]), 1.0)
                assert max(e_n, e_0, e_d) < 1e-8, (name, s, e_n, e_0, e_d)
                worst = max(worst, e_n, e_d)
                checked.append("golden %.1e/%.1e" % (e_n, e_d))
            else:
                # ratio between the first successful levels: rounding dependent, but bounded
                assert 0.01 <= mult[s] / gm[s] <= 100, (name, mult, gm)

                def single(log10_start, grad=True, stable=0):
                    ctx.set_option("start_mult_log10", log10_start)
                    ctx.set_option("stable", stable)
                    try:
                        return ctx.nll_batch(
                            KID[model["kernel"]], model["degree"], _lib.F64, hyp[s:s + 1, :pin["cov_N"]],
                            pin["m"][s:s + 1], pin["sn2"][s:s + 1], pin["vec"], grad,
                            None if pin["dm"] is None else pin["dm"][s:s + 1],
                            None if pin["dsn2"] is None else pin["dsn2"][s:s + 1])
                    finally:
                        ctx.set_option("start_mult_log10", 0)
                        ctx.set_option("stable", 0)

                lvl = int(round(np.log10(mult[s])))
                if lvl > 0:
                    # (i) the escalation path (failed samples gathered, re-run as a sub-batch IN STABLE MODE -- first at
                    # the level that failed, then x 10 per level -- results scattered back) must reproduce, bit for
                    # bit, a stable-mode run that STARTS at the level it ended on
                    n1, d1, m1, _, i1 = single(lvl, stable=1)
                    assert i1[0] == 0 and m1[0] == mult[s], (name, s, m1)
                    assert n1[0] == nlz[s] and np.array_equal(d1[0], dnlz[s], equal_nan=True), (name, s, "retry path")
                    checked.append("retry==start@1e%d" % lvl)
                x_n, x_d, cond, _, gsc = orc.core_extended(model, hyp[s], X, y, s2, sn2_mult=mult[s], with_scale=True)
                gsc = gsc * min(1.0, cond * EPS)  # see _errs
                bar = 8 * cond * EPS
                e_n, e_d = _errs(nlz[s], dnlz[s], x_n, x_d, gsc)
                if bar < 1e-2:
                    # (ii) digits exist at the device's level: value check against extended precision,
                    # and against the reference (golden / forced oracle) at twice the bar
                    bar = max(1e-8, bar)
                    assert e_n <= bar and e_d <= bar, (name, s, "vs extended precision", e_n, e_d, bar, cond)
                    checked.append("ext(cond %.0e) %.1e/%.1e bar %.1e" % (cond, e_n, e_d, bar))
                    if same:
                        r_n, r_d = _errs(nlz[s], dnlz[s], ref_n[s], ref_d[s], gsc)
                    else:
                        r_n, r_d = _errs(nlz[s], dnlz[s], *orc.core(model, hyp[s], X, y, s2, 1, 1, force_mult=mult[s]), gsc)
                    assert r_n <= 2 * bar and r_d <= 2 * bar, (name, s, "vs reference", r_n, r_d, bar)
                    checked.append("%s %.1e/%.1e" % ("golden" if same else "oracle@mult", r_n, r_d))
                else:
                    # (iii) cond * eps ~ 1 at the first successful level: no fp64 result has digits
                    # there (LAPACK's golden value is as far from the extended-precision one as the
                    # device's; printed).  nlZ is still within the bound; the VALUE check proper is
                    # made four decades up the same escalation ladder, where digits exist.
                    assert np.isfinite(nlz[s]) and np.isfinite(dnlz[s]).all() and e_n <= min(bar, 1.0), (name, s, e_n)
                    up = lvl + 4  # (five where four decades leave 8 cond eps just above 1e-2)
                    while True:
                        y_n, y_d, cond2, _, gsc2 = orc.core_extended(model, hyp[s], X, y, s2, sn2_mult=10.0 ** up, with_scale=True)
                        if 8 * cond2 * EPS < 1e-2 or up >= lvl + 6:
                            break
                        up += 1
                    n2, d2, m2, _, i2 = single(up)
                    assert i2[0] == 0 and m2[0] == 10.0 ** up, (name, s, m2)
                    gsc2 = gsc2 * min(1.0, cond2 * EPS)
                    bar2 = max(1e-8, 8 * cond2 * EPS)
                    u_n, u_d = _errs(n2[0], d2[0], y_n, y_d, gsc2)
                    assert bar2 < 1e-2 and u_n <= bar2 and u_d <= bar2, (name, s, "four decades up", u_n, u_d, bar2, cond2)
                    f_n, f_d = orc.core(model, hyp[s], X, y, s2, 1, 1, force_mult=10.0 ** up)
                    r_n, r_d = _errs(n2[0], d2[0], f_n, f_d, gsc2)
                    assert r_n <= 2 * bar2 and r_d <= 2 * bar2, (name, s, "vs oracle four decades up", r_n, r_d)
                    checked.append("first-success cond %.0e: nlZ err %.1e (LAPACK %.1e); @1e%d ext %.1e/%.1e oracle %.1e/%.1e bar %.1e"
                                   % (cond, e_n, abs(ref_n[s] - x_n) / max(1, abs(x_n)) if same else np.nan, up, u_n, u_d,
                                      r_n, r_d, bar2))
            assert checked, name
            report.append("%-44s s=%d mult %g (ref %g)  %s" % (name, s, mult[s], gm[s], "; ".join(checked)))
    print("\n".join(report))
    print("worst relative error over well-conditioned fixtures: %.3e" % worst)
    print("samples whose first successful jitter level differs from LAPACK's: %d of %d" %
          (n_diff_mult, 2 * len(g["names"])))
    print("  above LAPACK's level:", above)
    print("  below LAPACK's level:", below)
    # A failed factorization is retried at the SAME level in stable mode (refined panel solves: the accuracy of a
    # triangular solve) before the jitter goes up, so the device no longer needs systematically more jitter than the
    # reference (round 2: 1-2 decades more on 7 of the 8 singular samples, never less).  What is left is the coin
    # flip of a numerically indefinite matrix (tests/analysis/jitter_model.py: g029 at 10x has a negative fp64 eigenvalue).
    # DESIGN.md section 2 states the residue exactly: NO sample above LAPACK's level, and the two known samples
    # below it (g008 s0, g031 s1: 10 against 100).  The arithmetic is deterministic, so the test holds it to that.
    assert above == [], above
    assert set(b.split(":")[0] for b in below) <= {"g008 s=0", "g031 s=1"}, below


def test_arithmetic_after_a_jitter_retry(ctx, core_golden):
    """apply_mult, the sn2_mult factor of the noise gradient (:2497,:2503), sW and the predictive
    noise term with sn2_mult != 1, on WELL-conditioned systems where the comparison is sharp:
    the device starts its escalation at 10^3 (test hook) and the pinned oracle is evaluated with
    force_mult = 1000.  1e-8 relative on every plain fixture, posterior fields included."""
    from gpyreg_amd import _lib

    g = core_golden
    ctx.set_option("start_mult_log10", 3)
    try:
        for name in g["names"]:
            tag, model, N, D, flavour = parse_core_name(name)
            if flavour != "plain":
                continue
            X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
            s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
            pin = plugin_inputs(model, hyp, X, y, s2, True)
            ctx.set_data(X, y)
            nlz, dnlz, mult, lchol, info = ctx.nll_batch(
                KID[model["kernel"]], model["degree"], _lib.F64, hyp[:, :pin["cov_N"]], pin["m"], pin["sn2"],
                pin["vec"], True, pin["dm"], pin["dsn2"])
            assert (info == 0).all() and (mult == 1000).all(), (name, mult)
            post, pmult, _, pinfo = ctx.posterior_batch(
                KID[model["kernel"]], model["degree"], _lib.F64, hyp[:, :pin["cov_N"]], pin["m"], pin["sn2"],
                pin["vec"])
            assert (pinfo == 0).all() and (pmult == 1000).all()
            xs = g[tag + "_xs"]
            fmu, fs2 = post.predict(xs)
            for s in range(hyp.shape[0]):
                f_n, f_d = orc.core(model, hyp[s], X, y, s2, 1, 1, force_mult=1000)
                e_n, e_d = _errs(nlz[s], dnlz[s], f_n, f_d)
                assert e_n < 1e-8 and e_d < 1e-8, (name, s, e_n, e_d)
                ref = orc.core(model, hyp[s], X, y, s2, 0, 0, force_mult=1000)
                alpha, sW, L = post.fetch(s)
                assert np.abs(alpha - ref.alpha[:, 0]).max() <= 1e-8 * np.abs(ref.alpha).max(), name
                assert np.allclose(sW, ref.sW[:, 0], rtol=1e-12), name
                mine = L.T if lchol[s] else L
                assert np.abs(mine - ref.L).max() <= 1e-8 * np.abs(ref.L).max(), name
                rmu, rs2 = orc.predict(model, [ref], X, y, xs, separate_samples=True)
                cov_N, noise_N = pin["cov_N"], orc.noise_count(model["noise"])
                m_star = orc.mean(model["mean"], hyp[s, cov_N + noise_N:], xs)
                assert np.abs(m_star + fmu[:, s] - rmu[:, 0]).max() <= 1e-8 * max(1.0, np.abs(rmu).max()), name
                sf2 = np.exp(2 * hyp[s, cov_N - 1 if model["kernel"] != "rq" else cov_N - 2])
                assert np.abs(np.maximum(fs2[:, s], 0) - rs2[:, 0]).max() <= 1e-8 * sf2, name
            post.free()
    finally:
        ctx.set_option("start_mult_log10", 0)


def test_fp32_mode_against_the_reference_goldens(ctx, core_golden):
    """north_star: fp32 within 1e-3 relative of the NumPy/SciPy reference.  Every well-conditioned
    fixture (cond(A) * eps_32 < 1e-3 / 8, i.e. every "plain" one) against the reference's golden
    nlZ / dnlZ / predictions, fp32 factorization, through the C ABI."""
    from gpyreg_amd import _lib

    g = core_golden
    worst, n = 0.0, 0
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        if flavour != "plain":
            continue
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        pin = plugin_inputs(model, hyp, X, y, s2, True)
        ctx.set_data(X, y)
        nlz, dnlz, mult, lchol, info = ctx.nll_batch(
            KID[model["kernel"]], model["degree"], _lib.F32, hyp[:, :pin["cov_N"]], pin["m"], pin["sn2"],
            pin["vec"], True, pin["dm"], pin["dsn2"])
        assert (info == 0).all() and np.array_equal(mult, g[tag + "_sn2_mult"]), name
        post, *_ = ctx.posterior_batch(KID[model["kernel"]], model["degree"], _lib.F32, hyp[:, :pin["cov_N"]],
                                       pin["m"], pin["sn2"], pin["vec"])
        xs = g[tag + "_xs"]
        fmu, fs2 = post.predict(xs)
        post.free()
        cov_N, noise_N = pin["cov_N"], orc.noise_count(model["noise"])
        for s in range(hyp.shape[0]):
            e_n, e_d = _errs(nlz[s], dnlz[s], g[tag + "_nlZ"][s], g[tag + "_dnlZ"][s])
            assert e_n < 1e-3 and e_d < 1e-3, (name, s, e_n, e_d)
            worst, n = max(worst, e_n, e_d), n + 1
            m_star = orc.mean(model["mean"], hyp[s, cov_N + noise_N:], xs)
            rm, rv = g[tag + "_mu_sep"][:, s], g[tag + "_s2_sep"][:, s]
            sf2 = np.exp(2 * hyp[s, cov_N - 1 if model["kernel"] != "rq" else cov_N - 2])
            assert np.abs(m_star + fmu[:, s] - rm).max() <= 1e-3 * max(1.0, np.abs(rm).max()), name
            assert np.abs(np.maximum(fs2[:, s], 0) - rv).max() <= 1e-3 * sf2, name
    print("fp32 vs reference goldens: worst relative error %.2e over %d samples" % (worst, n))


def test_posterior_and_predict_match_golden(ctx, core_golden):
    from gpyreg_amd import _lib

    g = core_golden
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        pin = plugin_inputs(model, hyp, X, y, s2, False)
        ctx.set_data(X, y)
        post, mult, lchol, info = ctx.posterior_batch(
            KID[model["kernel"]], model["degree"], _lib.F64, hyp[:, :pin["cov_N"]], pin["m"], pin["sn2"],
            pin["vec"])
        assert (info == 0).all(), name
        assert np.array_equal(lchol, g[tag + "_L_chol"]), name
        same_mult = np.array_equal(mult, g[tag + "_sn2_mult"])
        plain = flavour == "plain"
        if plain:
            assert same_mult, name
        xs = g[tag + "_xs"]
        fmu, fs2 = post.predict(xs)
        cov_N = pin["cov_N"]
        noise_N = orc.noise_count(model["noise"])
        # ill-conditioned fixtures: alpha, L and the predictions carry an error of order cond * eps;
        # they are compared with the pinned oracle at the DEVICE's multiplier at that bar (capped:
        # beyond cond ~ 1e13 only finiteness and the flags are meaningful, and the NLL test above
        # has already value-checked the factorization against extended precision)
        if not plain:
            for s in range(hyp.shape[0]):
                cond = orc.core_extended(model, hyp[s], X, y, s2, sn2_mult=mult[s])[2]
                alpha, sW, L = post.fetch(s)
                assert np.isfinite(alpha).all() and np.isfinite(L).all() and np.isfinite(fmu).all(), name
                try:
                    ref = orc.core(model, hyp[s], X, y, s2, 0, 0, force_mult=mult[s])
                except np.linalg.LinAlgError:
                    continue
                assert np.allclose(sW, ref.sW[:, 0], rtol=1e-12), (name, "sW")
                bar = 50 * cond * EPS
                if bar < 1e-3:
                    assert np.abs(alpha - ref.alpha[:, 0]).max() <= max(1e-8, bar) * np.abs(ref.alpha).max(), (name, "alpha")
                    mine = L.T if lchol[s] else L
                    assert np.abs(mine - ref.L).max() <= max(1e-8, bar) * np.abs(ref.L).max(), (name, "L")
            post.free()
            continue
        tol = 1e-8
        for s in range(hyp.shape[0]):
            alpha, sW, L = post.fetch(s)
            ra = g[tag + "_alpha"][s]
            assert np.abs(alpha - ra).max() <= tol * np.abs(ra).max(), (name, "alpha")
            assert np.allclose(sW, g[tag + "_sW"][s], rtol=1e-12), (name, "sW")
            # reference Posterior.L is the UPPER factor (or -inv): ours is its transpose / same
            Lref = None
            if tag + "_L" in g.files:
                Lref = g[tag + "_L"][s]
                mine = L.T if lchol[s] else L
                assert np.abs(mine - Lref).max() <= tol * np.abs(Lref).max(), (name, "L")
            else:
                mine = L.T if lchol[s] else L
                assert np.abs(np.diag(mine) - g[tag + "_Ldiag"][s]).max() <= tol * np.abs(g[tag + "_Ldiag"][s]).max()
                assert np.abs(mine[0] - g[tag + "_Lrow0"][s]).max() <= tol * np.abs(g[tag + "_Lrow0"][s]).max()
                assert np.abs(mine[:, -1] - g[tag + "_Lcol_last"][s]).max() <= tol * np.abs(g[tag + "_Lcol_last"][s]).max()
                assert abs(np.linalg.norm(mine) - g[tag + "_Lfro"][s]) <= tol * g[tag + "_Lfro"][s]
            m_star = orc.mean(model["mean"], hyp[s, cov_N + noise_N:], xs)
            mu = m_star + fmu[:, s]
            v = np.maximum(fs2[:, s], 0)
            rm, rv = g[tag + "_mu_sep"][:, s], g[tag + "_s2_sep"][:, s]
            assert np.abs(mu - rm).max() <= tol * max(1.0, np.abs(rm).max()), (name, "mu")
            # predictive variances are differences of O(sf2) numbers: absolute scale sf2
            sf2 = np.exp(2 * hyp[s, cov_N - 1 if model["kernel"] != "rq" else cov_N - 2])
            assert np.abs(v - rv).max() <= 1e-7 * sf2, (name, "s2")
        post.free()

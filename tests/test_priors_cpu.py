"""CPU parity of the host-side prior / design code against reference-generated fixtures
(tests/golden/prior_cases.npz): log-prior values and gradients, truncation constants, and
the f_min_fill design under a fixed global NumPy seed."""

import numpy as np
import pytest

from gpyreg_amd import priors as pr
from gpyreg_amd.f_min_fill import f_min_fill


@pytest.fixture(scope="module")
def g():
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", "prior_cases.npz"), allow_pickle=False)


def _reference_broadcast_is_benign(hp, lb, ub, h):
    """The reference forms the smooth-box constant C over ALL smooth-box dimensions and
    broadcasts it against the subset outside (or inside) [a, b] (gaussian_process.py:
    1330-1356, :1371-1413).  That is only the intended per-dimension formula when a class
    has a single dimension or all of its dimensions fall in the same region; otherwise it
    double counts (or raises).  gpyreg_amd computes the per-dimension value, so rows where
    the reference's broadcast is not benign are not comparable."""
    ix = pr.classify(hp, lb, ub)
    for cls in ("sb", "sb_t"):
        sel = ix[cls]
        if sel.sum() <= 1:
            continue
        out = ((h < hp["a"]) | (h > hp["b"])) & sel
        if 0 < out.sum() < sel.sum():
            return False
    return True


def test_log_priors_and_normalization_match_reference(g):
    compared = 0
    for name in g["names"]:
        tag = str(name).split("|")[0]
        hp = {k: g[tag + "_" + k] for k in ("mu", "sigma", "df", "a", "b")}
        lb, ub = g[tag + "_lb"], g[tag + "_ub"]
        norm = pr.normalization_constants(hp, lb, ub)
        assert np.allclose(norm, g[tag + "_norm"], rtol=1e-14, atol=0), tag
        for r, h in enumerate(g[tag + "_H"]):
            lp, dlp = pr.log_priors(h, hp, lb, ub, norm, True)
            ref_lp, ref_d = g[tag + "_lp"][r], g[tag + "_dlp"][r]
            assert np.allclose(dlp, ref_d, rtol=1e-12, atol=1e-14, equal_nan=True), (tag, r)
            if not _reference_broadcast_is_benign(hp, lb, ub, h):
                continue
            compared += 1
            if np.isfinite(ref_lp):
                assert abs(lp - ref_lp) <= 1e-12 * max(1.0, abs(ref_lp)), (tag, r)
            else:
                assert lp == ref_lp, (tag, r)
            assert np.allclose(dlp, ref_d, rtol=1e-12, atol=1e-14, equal_nan=True), (tag, r)
            assert pr.log_priors(h, hp, lb, ub, norm, False) == lp
    assert compared >= 40


def test_multi_dimensional_smoothbox_reproduces_the_reference_arithmetic(g):
    """Smooth-box (and smooth-box Student-t) priors on several dimensions that fall in different
    regions: the reference broadcasts the per-class normaliser against the subset outside / inside
    the box (gaussian_process.py:1346-1356, :1391-1413).  With two dimensions that counts terms
    twice; with three it raises numpy's broadcasting ValueError.  Both are reproduced: values
    bit-exact, and the exception where the reference raises."""
    from gpyreg_amd import priors as pr

    for tag in ("q0", "q1", "q2", "q3"):
        hp = {k: g[tag + "_" + k] for k in ("mu", "sigma", "df", "a", "b")}
        lb, ub, norm = g[tag + "_lb"], g[tag + "_ub"], g[tag + "_norm"]
        assert np.allclose(pr.normalization_constants(hp, lb, ub), norm, rtol=1e-12, equal_nan=True)
        H, lp, dlp, raised = g[tag + "_H"], g[tag + "_lp"], g[tag + "_dlp"], g[tag + "_raised"]
        assert raised.any() == (tag in ("q0", "q1"))
        for r in range(H.shape[0]):
            if raised[r]:
                with pytest.raises(ValueError):
                    pr.log_priors(H[r], hp, lb, ub, norm, True)
            else:
                a, b = pr.log_priors(H[r], hp, lb, ub, norm, True)
                assert a == lp[r] and np.array_equal(b, dlp[r], equal_nan=True), (tag, r, a, lp[r])


def test_design_matches_reference_under_same_seed(g):
    for idx, name in enumerate(g["names"]):
        tag = str(name).split("|")[0]
        hp = {k: g[tag + "_" + k] for k in ("mu", "sigma", "df", "a", "b")}
        np.random.seed(4321 + idx)
        fb = lambda X: np.array([np.sum((h - 0.3) ** 2) + np.sin(3 * h[0]) for h in X])
        X, y = f_min_fill(fb, g[tag + "_dx0"], g[tag + "_dLB"], g[tag + "_dUB"], g[tag + "_dPLB"],
                          g[tag + "_dPUB"], hp, 40, "sobol")
        assert np.allclose(X, g[tag + "_dX"], rtol=1e-13, atol=1e-13), tag
        assert np.allclose(y, g[tag + "_dy"], rtol=1e-13), tag


def test_smoothbox_cdf_ppf_roundtrip_and_uuinv():
    for sigma, a, b in [(0.7, -1.0, 2.0), (2.0, 0.0, 0.5)]:
        for q in [0.01, 0.2, 0.5, 0.77, 0.99]:
            assert abs(pr.smoothbox_cdf(pr.smoothbox_ppf(q, sigma, a, b), sigma, a, b) - q) < 1e-12
            x = pr.smoothbox_student_t_ppf(q, 4.0, sigma, a, b)
            assert abs(pr.smoothbox_student_t_cdf(x, 4.0, sigma, a, b) - q) < 1e-10
    p = np.linspace(0, 1, 11)
    x = pr.uuinv(p, [-3.0, -1.0, 1.0, 3.0], 0.5)
    assert np.all(np.diff(x) > 0) and x[0] == -3.0 and abs(x[-1] - 3.0) < 1e-12
    assert np.allclose(pr.uuinv(p, [0, 0, 1, 1], 0.5)[[0, -1]], [0, 1])


def test_slice_sampler_moments():
    """Standard normal in a wide box: mean ~0, variance ~1 (statistical, seeded)."""
    from gpyreg_amd.slice_sample import SliceSampler

    np.random.seed(7)
    s = SliceSampler(lambda x: -0.5 * np.sum(x**2), np.array([0.5, -0.5]), np.array([1.0, 1.0]),
                     np.array([-10.0, -10.0]), np.array([10.0, 10.0]))
    res = s.sample(4000, burn=300)
    xs = res["samples"]
    assert np.abs(xs.mean(0)).max() < 0.1 and np.abs(xs.var(0) - 1).max() < 0.15
    assert xs.min() > -10 and xs.max() < 10
    # bounds respected when the mode is outside the box
    np.random.seed(8)
    s = SliceSampler(lambda x: -0.5 * np.sum((x - 3) ** 2), np.array([0.5]), None, np.array([0.0]), np.array([1.0]))
    xs = s.sample(500)["samples"]
    assert xs.min() >= 0 and xs.max() <= 1 and xs.mean() > 0.55


def test_speculative_slice_sampler_walks_the_sequential_chain():
    """The shrinkage proposals of a coordinate update do not depend on the target's values, only the point at which
    they stop does: evaluated four at a time (one device batch), with the global RNG rewound to the draws the
    sequential procedure makes, the sampler must return the same samples, values, adapted widths and leave the
    same RNG stream behind -- with fewer calls.  A batch that raises hands the coordinate back to the sequential loop."""
    from gpyreg_amd.slice_sample import SliceSampler

    scale = np.array([1.0, 0.3, 2.0])

    def logf(x):
        return float(-0.5 * np.sum((x / scale) ** 2) + np.log1p(0.5 * np.cos(3 * x[0]) ** 2))

    calls = {"n": 0}

    def logf_batch(X):
        calls["n"] += 1
        if calls["n"] % 7 == 0:
            raise np.linalg.LinAlgError("a speculative row failed")
        return np.array([logf(x) for x in X])

    for step_out in (False, True):  # (with step_out the interval ends of the coordinates already updated in a sweep
        out = []                    # are read again: the batched path must leave them where the sequential one does)
        for batch in (None, logf_batch):
            calls["n"] = 0
            np.random.seed(5)
            s = SliceSampler(logf, np.zeros(3), None, -4 * np.ones(3), 4 * np.ones(3),
                             {"log_f_batch": batch, "speculate": 4, "step_out": step_out})
            r = s.sample(120, thin=2, burn=40)
            out.append((r["samples"], r["f_vals"], s.widths.copy(), np.random.rand(), s.func_count, s.device_calls))
        (sa, fa, wa, ra, na, da), (sb, fb, wb, rb, nb, db) = out
        assert np.array_equal(sa, sb) and np.array_equal(fa, fb) and np.array_equal(wa, wb) and ra == rb, step_out
        assert na == nb and db < (0.9 if step_out else 0.75) * da


def test_log_priors_rows_is_bit_identical_to_the_row_loop(g):
    """The vectorised prior evaluation of the design stage against ``log_priors`` row by row: the reference-pinned
    configurations of prior_cases.npz, and random mixtures of all six classes (one smooth-box dimension of each kind,
    so that the vectorised branch runs; several of them take the row loop by design)."""
    cases = []
    for name in g["names"]:
        tag = str(name).split("|")[0]
        hp = {k: g[tag + "_" + k] for k in ("mu", "sigma", "df", "a", "b")}
        cases.append((hp, g[tag + "_lb"], g[tag + "_ub"], g[tag + "_H"]))
    rng = np.random.default_rng(11)
    for trial in range(30):
        D = 9
        hp = pr.empty_priors(D)
        kinds = rng.permutation(["gauss", "gauss", "stud", "stud", "sb", "sb_t", "uni", "fixed", "gauss"])
        lb, ub = np.full(D, -8.0), np.full(D, 8.0)
        for i, k in enumerate(kinds):
            if k == "gauss":
                hp["mu"][i], hp["sigma"][i], hp["df"][i] = rng.normal(), 0.3 + rng.random(), 0
            elif k == "stud":
                hp["mu"][i], hp["sigma"][i], hp["df"][i] = rng.normal(), 0.3 + rng.random(), rng.integers(1, 8)
            elif k == "sb":
                hp["a"][i], hp["b"][i], hp["sigma"][i], hp["df"][i] = -1 - rng.random(), 1 + rng.random(), 0.5 + rng.random(), 0
            elif k == "sb_t":
                hp["a"][i], hp["b"][i], hp["sigma"][i], hp["df"][i] = -1 - rng.random(), 1 + rng.random(), 0.5 + rng.random(), 3
            elif k == "fixed":
                lb[i] = ub[i] = 0.25
        H = rng.uniform(-4, 4, (40, D))
        H[:, kinds == "fixed"] = 0.25
        H[3, kinds == "fixed"] = 0.3  # off the fixed value: -inf
        H[5, 0] = np.nan
        cases.append((hp, lb, ub, H))
    compared = 0
    for hp, lb, ub, H in cases:
        norm = pr.normalization_constants(hp, lb, ub)
        try:
            rows = [pr.log_priors(h, hp, lb, ub, norm, True) for h in H]
        except ValueError:  # the reference's broadcasting error (three smooth-box dimensions): the same on both paths
            with pytest.raises(ValueError):
                pr.log_priors_rows(H, hp, lb, ub, norm, True)
            continue
        lp, dlp = pr.log_priors_rows(H, hp, lb, ub, norm, True)
        lp0, none = pr.log_priors_rows(H, hp, lb, ub, norm, False)
        assert none is None
        for r, (v, d) in enumerate(rows):
            assert np.array_equal(np.float64(v), lp[r], equal_nan=True) and np.array_equal(d, dlp[r], equal_nan=True)
            assert np.array_equal(lp0[r], lp[r], equal_nan=True)
            compared += 1
    assert compared >= 1000


def test_slice_sampler_sweep_matches_the_reference_output():
    """tools/sampler_sweep.py (a correlated Gaussian, a bounded skewed density; step_out / adaptive / caller's widths /
    bounds / thinning / burn-in / a second call continuing the chain; the error messages) against the output of the
    reference's SliceSampler under the same seeds: samples, function values and the state of NumPy's global generator
    afterwards, to twelve digits."""
    import contextlib
    import io
    import os
    import runpy
    import warnings

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    buf = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stdout(buf):
        warnings.simplefilter("ignore")
        runpy.run_path(os.path.join(root, "tools", "sampler_sweep.py"), run_name="__main__")
    mine = buf.getvalue().splitlines()
    with open(os.path.join(root, "tests", "golden", "sampler_sweep_reference.txt")) as f:
        ref = [ln.rstrip("\n") for ln in f]
    assert len(ref) == 51 and mine == ref, [(r, m) for r, m in zip(ref, mine) if r != m][:3]

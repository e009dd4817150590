"""Pins the CPU oracle (oracle/gp_oracle.py) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only.

The oracle calls the same SciPy/NumPy entry points in the same order as the
reference, so the comparison is bit-exact (``np.array_equal`` with NaNs equal).
"""

import numpy as np
import pytest

from conftest import parse_core_name, parse_cov_name
from oracle import gp_oracle as orc


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_cov_golden_bit_exact(cov_golden):
    g = cov_golden
    assert len(g["names"]) == 27
    for name in g["names"]:
        tag, kernel, degree, N, D, M = parse_cov_name(name)
        X, Xs, hyp = g[tag + "_X"], g[tag + "_Xs"], g[tag + "_hyp"]
        K, dK = orc.covariance(kernel, hyp, X, compute_grad=True, degree=degree)
        assert _eq(K, g[tag + "_K"]), name
        assert _eq(dK, g[tag + "_dK"]), name
        assert _eq(orc.covariance(kernel, hyp, X, Xs, degree=degree), g[tag + "_Ks"]), name
        assert _eq(
            orc.covariance(kernel, hyp, Xs, compute_diag=True, degree=degree), g[tag + "_kd"]
        ), name


def test_core_golden_bit_exact(core_golden):
    g = core_golden
    assert len(g["names"]) == 35
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        for s in range(hyp.shape[0]):
            nlZ, dnlZ = orc.core(model, hyp[s], X, y, s2, 1, 1)
            assert _eq(nlZ, g[tag + "_nlZ"][s]), name
            assert _eq(dnlZ, g[tag + "_dnlZ"][s]), name
            assert _eq(orc.core(model, hyp[s], X, y, s2, 1, 0), g[tag + "_nlZ_only"][s]), name
        posts = orc.posteriors(model, hyp, X, y, s2)
        for s, p in enumerate(posts):
            assert _eq(p.alpha[:, 0], g[tag + "_alpha"][s]), name
            assert _eq(p.sW[:, 0], g[tag + "_sW"][s]), name
            assert float(p.sn2_mult) == g[tag + "_sn2_mult"][s], name
            assert bool(p.L_chol) == bool(g[tag + "_L_chol"][s]), name
            if tag + "_L" in g.files:
                assert _eq(p.L, g[tag + "_L"][s]), name
            else:
                assert _eq(np.diag(p.L), g[tag + "_Ldiag"][s]), name
                assert _eq(np.asarray(p.L)[0], g[tag + "_Lrow0"][s]), name
                assert _eq(np.asarray(p.L)[:, -1], g[tag + "_Lcol_last"][s]), name
        xs, ys = g[tag + "_xs"], g[tag + "_ys"]
        s2s = 0.02 * np.ones((xs.shape[0], 1)) if s2 is not None else None
        mu, v = orc.predict(model, posts, X, y, xs, ys, s2s, separate_samples=True)
        assert _eq(mu, g[tag + "_mu_sep"]) and _eq(v, g[tag + "_s2_sep"]), name
        mu, v, lpd = orc.predict(model, posts, X, y, xs, ys, s2s, add_noise=True, return_lpd=True)
        assert _eq(mu, g[tag + "_mu_avg"]) and _eq(v, g[tag + "_s2n_avg"]), name
        assert _eq(lpd, g[tag + "_lpd_avg"]), name
        _, _, lpd = orc.predict(
            model, posts, X, y, xs, ys, s2s, separate_samples=True, return_lpd=True
        )
        assert _eq(lpd, g[tag + "_lpd_sep"]), name


def test_golden_covers_edge_semantics(core_golden):
    """The fixtures really contain the edge cases SURVEY section 8a lists."""
    g = core_golden
    names = [str(n) for n in g["names"]]
    # jitter escalation observed in the reference
    assert any(g[n.split("|")[0] + "_sn2_mult"].max() > 1 for n in names)
    # low-noise branch (L_chol False)
    assert any(not g[n.split("|")[0] + "_L_chol"].all() for n in names)
    # Matern-1 NaN lengthscale gradients, finite elsewhere
    tag = [n for n in names if "|matern1|" in n][0].split("|")[0]
    d = g[tag + "_dnlZ"]
    assert np.isnan(d[:, :2]).all() and np.isfinite(d[:, 2:]).all()


def test_synthetic_problem_matches_survey_value():
    """reference nlZ(s=0) for cfg2 is -1166.298896135079."""
    model, X, y, hyp = orc.synthetic_problem(2)
    nlZ = orc.core(model, hyp[0], X, y, None, 1, 0)
    assert abs(nlZ - (-1166.298896135079)) < 1e-6 * 1166


def test_force_mult_and_extended_precision_restatements(core_golden):
    """The two test-only variants of the oracle: ``force_mult`` at the reference's own final
    multiplier reproduces the golden values bit for bit, and the extended-precision evaluation
    agrees with the reference to a small multiple of cond(A) * eps on every fixture it is used for
    (that multiple, 0.6 here, is what the GPU tests' bar of 8 is judged against)."""
    g = core_golden
    worst = 0.0
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        if N > 140:
            continue
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        for s in range(hyp.shape[0]):
            m = g[tag + "_sn2_mult"][s]
            rn, rd = g[tag + "_nlZ"][s], g[tag + "_dnlZ"][s]
            ok = ~np.isnan(rd)
            n2, d2 = orc.core(model, hyp[s], X, y, s2, 1, 1, force_mult=m)
            assert n2 == rn and np.array_equal(d2[ok], rd[ok]), name
            xn, xd, cond, lchol = orc.core_extended(model, hyp[s], X, y, s2, sn2_mult=m)
            assert lchol == bool(g[tag + "_L_chol"][s])
            e_n = abs(xn - rn) / max(1.0, abs(rn))
            e_d = np.max(np.abs(xd[ok] - rd[ok]) / np.maximum(np.abs(rd[ok]), np.abs(rd[ok]).max()))
            ratio = max(e_n, e_d) / (cond * np.finfo(float).eps)
            assert max(e_n, e_d) < 1e-10 or ratio < 2.0, (name, s, e_n, e_d, cond)
            if max(e_n, e_d) >= 1e-10:
                worst = max(worst, ratio)
    assert worst < 2.0
    # one level below LAPACK's first success the forced oracle must refuse, not fall through
    tag = [n.split("|")[0] for n in g["names"] if "jitter_low" in n][0]
    model = parse_core_name([n for n in g["names"] if n.startswith(tag)][0])[1]
    with pytest.raises(np.linalg.LinAlgError):
        orc.core(model, g[tag + "_hyp"][0], g[tag + "_X"], g[tag + "_y"], None, 1, 1,
                 force_mult=g[tag + "_sn2_mult"][0] / 100)


def test_rank_one_update_bit_exact():
    """The oracle's restatement of the reference's rank-one update path (gaussian_process.py:750-844)
    against the reference's own results (rank1_cases.npz): three consecutive one-point updates,
    high- and low-noise parametrisation."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rank1_cases.npz"), allow_pickle=False)
    for name in g["names"]:
        tag, kname, mname, npar, N, D, flav = str(name).split("|")
        degree, kernel = 0, kname
        if kname.startswith("matern"):
            kernel, degree = "matern", int(kname[6:])
        model = dict(kernel=kernel, degree=degree, mean=mname, noise=tuple(int(c) for c in npar))
        X, y, hyp, xs = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"], g[tag + "_xs"]
        posts = orc.posteriors(model, hyp, X, y, None)
        for k in range(3):
            posts, X, y, full = orc.rank_one_update(model, posts, X, y, g[tag + "_Xn"][k:k + 1], g[tag + "_yn"][k:k + 1])
            assert full == []
            mu, s2 = orc.predict(model, posts, X, y, xs, separate_samples=True)
            assert _eq(mu, g[tag + f"_mu{k}"]) and _eq(s2, g[tag + f"_s2{k}"]), (name, k)
        for s, p in enumerate(posts):
            assert _eq(p.alpha[:, 0], g[tag + "_alpha"][s]) and _eq(p.sW[:, 0], g[tag + "_sW"][s]), name
            assert _eq(np.diag(p.L), g[tag + "_Ldiag"][s]) and _eq(np.asarray(p.L)[:, -1], g[tag + "_Llast_col"][s]), name
            assert _eq(np.asarray(p.L)[-1, :], g[tag + "_Llast_row"][s]), name


def test_streamed_core_is_the_core_bit_for_bit(cov_golden, core_golden):
    """``core_streamed`` (one gradient plane at a time -- the only way to a gradient at cfg4's size) against the
    reference's own values: every plane of every cov_cases.npz kernel, and nlZ / dnlZ of all 35 core_cases.npz
    models, with ``np.array_equal``.  Bit-exact, not 1e-13, because the planes are produced by the same SciPy calls
    on the same operands (only loop-invariant factors are hoisted) and contracted from a contiguous (N, N) array,
    which is what ``dK[:, :, i]`` of the reference's (cov_N, N, N) C array is."""
    g = cov_golden
    for name in g["names"]:
        tag, kernel, degree, N, D, M = parse_cov_name(name)
        planes = list(orc.covariance_planes(kernel, g[tag + "_hyp"], g[tag + "_X"], degree=degree))
        assert _eq(planes[0], g[tag + "_K"]), name
        dK = g[tag + "_dK"]
        assert len(planes) == dK.shape[2] + 1
        for i in range(dK.shape[2]):
            assert planes[1 + i].flags.c_contiguous and _eq(planes[1 + i], dK[:, :, i]), (name, i)
    g = core_golden
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        for s in range(hyp.shape[0]):
            nlZ, dnlZ = orc.core_streamed(model, hyp[s], X, y, s2)
            assert _eq(nlZ, g[tag + "_nlZ"][s]) and _eq(dnlZ, g[tag + "_dnlZ"][s]), name


def test_streamed_core_at_cfg3_full_size_and_the_cfg4_fixture():
    """cfg3 sample 0 at N = 4096 (fullsize_cases.npz: the reference's own nlZ / dnlZ): the streamed oracle gives the
    same bits (~20 s).  And the provenance of cfg4's gradient fixture: labelled oracle-derived (the reference would
    need 47 GB), produced by tests/golden/make_golden.py cfg4grad, which asserted that the streamed oracle's nlZ at
    cfg4 IS the reference's stored nlZ (bit for bit) before it wrote the gradient."""
    import os

    here = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(here, "fullsize_cases.npz"), allow_pickle=False)
    model, X, y, hyp = orc.synthetic_problem(3, S=16)
    assert np.array_equal(hyp[0], g["cfg3_hyp"][0])
    nlZ, dnlZ = orc.core_streamed(model, hyp[0], X, y, None)
    assert nlZ == g["cfg3_nlZ"][0] and np.array_equal(dnlZ, g["cfg3_dnlZ"][0])
    g4 = np.load(os.path.join(here, "fullsize45_cases.npz"), allow_pickle=False)
    assert g4["cfg4_dnlZ"].shape == (1, 24) and np.isfinite(g4["cfg4_dnlZ"]).all()
    assert "oracle-derived" in str(g4["cfg4_dnlZ_source"])

"""Hyperparameter priors: densities, truncation constants and the helper distributions
used to draw the initial design (host-side scalar/vector math, O(hyp_N)).

Restates, for the reference's four prior families,
  GP.__recompute_normalization_constants   gaussian_process.py:1234-1273
  GP.__compute_log_priors                  gaussian_process.py:1275-1466
  smoothbox / smoothbox-Student-t cdf, ppf f_min_fill.py:249-372
  uuinv (mixture-of-uniforms inverse cdf)  f_min_fill.py:183-246
The value added to the negative log marginal likelihood must be identical to the
reference's (the optimiser and the sampler see nlZ - log prior), including the
reference's classification rule: its test ``df == 0 | ~isfinite(df)`` parses as
``df == (0 | ~isfinite(df))`` and is true exactly when ``df == 0`` (a finite df of 1
compares against 0, a non-finite df against 1), so "Gaussian-like" means df == 0.
"""

from __future__ import annotations

import numpy as np
import scipy.special as sps
import scipy.stats as sst

_S2PI = np.sqrt(2 * np.pi)


def _t_norm(df, sigma):
    """Peak density of a scaled Student-t: Gamma((df+1)/2) / (Gamma(df/2) sigma sqrt(df pi))."""
    return sps.gamma(0.5 * (df + 1)) / (sps.gamma(0.5 * df) * sigma * np.sqrt(df * np.pi))


# ---- smooth box: flat on [a, b], Gaussian (or Student-t) shoulders of scale sigma ----
def smoothbox_cdf(x, sigma, a, b):
    C = 1.0 + (b - a) / (sigma * _S2PI)
    if x < a:
        return sst.norm.cdf(x, loc=a, scale=sigma) / C
    if x <= b:
        return (0.5 + (x - a) / (sigma * _S2PI)) / C
    return (C - 1.0 + sst.norm.cdf(x, loc=b, scale=sigma)) / C


def smoothbox_ppf(q, sigma, a, b):
    C = 1.0 + (b - a) / (sigma * _S2PI)
    if q < 0.5 / C:
        return sst.norm.ppf(C * q, loc=a, scale=sigma)
    if q <= (C - 0.5) / C:
        return (q * C - 0.5) * sigma * _S2PI + a
    return sst.norm.ppf(C * q - (C - 1), loc=b, scale=sigma)


def smoothbox_student_t_cdf(x, df, sigma, a, b):
    c = _t_norm(df, sigma)
    C = 1.0 + (b - a) * c
    if x < a:
        return sst.t.cdf(x, df, loc=a, scale=sigma) / C
    if x <= b:
        return (0.5 + (x - a) * c) / C
    return (C - 1.0 + sst.t.cdf(x, df, loc=b, scale=sigma)) / C


def smoothbox_student_t_ppf(q, df, sigma, a, b):
    c = _t_norm(df, sigma)
    C = 1.0 + (b - a) * c
    if q < 0.5 / C:
        return sst.t.ppf(C * q, df, loc=a, scale=sigma)
    if q <= (C - 0.5) / C:
        return (q * C - 0.5) / c + a
    return sst.t.ppf(C * q - (C - 1), df, loc=b, scale=sigma)


def uuinv(p, B, w):
    """Inverse cdf of  w U(B1,B2) + (1-w)/2 [U(B0,B1) + U(B2,B3)]  (B = LB, PLB, PUB, UB)."""
    assert B[0] <= B[1] <= B[2] <= B[3]
    assert 0 <= w <= 1
    p = np.asarray(p, dtype=float)
    x = np.zeros(p.shape)
    L = B[3] - B[0] + B[1] - B[2]  # total length of the two outer segments
    if w == 1:
        return p * (B[2] - B[1]) + B[1]
    if L == 0:  # outer segments collapse to point masses
        lo = p <= (1 - w) / 2
        x[lo] = B[0]
        if w != 0:
            mid = (p <= (1 - w) / 2 + w) & ~lo
            x[mid] = (p[mid] - (1 - w) / 2) * (B[2] - B[1]) / w + B[1]
        x[p > (1 - w) / 2 + w] = B[3]
        return x
    t1 = (1 - w) * (B[1] - B[0]) / L  # mass left of PLB
    lo = p <= t1
    x[lo] = B[0] + p[lo] * L / (1 - w)
    mid = (p <= t1 + w) & ~lo
    if w != 0:
        x[mid] = (p[mid] - t1) * (B[2] - B[1]) / w + B[1]
    hi = p > t1 + w
    x[hi] = (p[hi] - w - t1) * L / (1 - w) + B[2]
    x[p < 0] = np.nan
    x[p > 1] = np.nan
    return x


# ---- prior bookkeeping ---------------------------------------------------------------
def empty_priors(hyp_N):
    return {k: np.full((hyp_N,), np.nan) for k in ("mu", "sigma", "df", "a", "b")}


def normalization_constants(hp, lb, ub):
    """Mass of each (untruncated) prior inside [lb, ub] (gaussian_process.py:1234-1273)."""
    out = np.full(np.shape(lb), 1.0)
    for i in range(np.size(lb)):
        mu, sigma, df = hp["mu"][i], np.abs(hp["sigma"])[i], hp["df"][i]
        a, b = hp["a"][i], hp["b"][i]
        if lb[i] == ub[i]:
            continue  # fixed dimension
        if not np.isfinite(lb[i]) and not np.isfinite(ub[i]):
            continue  # unbounded
        if not np.isfinite(mu) and not np.isfinite(sigma):
            continue  # uniform
        gaussian_like = df == 0 or not np.isfinite(df)
        if np.isfinite(a) and np.isfinite(b):
            if gaussian_like:
                lo, hi = smoothbox_cdf(lb[i], sigma, a, b), smoothbox_cdf(ub[i], sigma, a, b)
            else:
                lo = smoothbox_student_t_cdf(lb[i], df, sigma, a, b)
                hi = smoothbox_student_t_cdf(ub[i], df, sigma, a, b)
        elif gaussian_like:
            lo, hi = sst.norm.cdf(lb[i], loc=mu, scale=sigma), sst.norm.cdf(ub[i], loc=mu, scale=sigma)
        else:
            lo, hi = sst.t.cdf(lb[i], df, loc=mu, scale=sigma), sst.t.cdf(ub[i], df, loc=mu, scale=sigma)
        out[i] = hi - lo
    return out


def classify(hp, lb, ub):
    """Index sets of gaussian_process.py:1289-1312 (see module docstring for ``df == 0``)."""
    mu, sigma, df, a, b = hp["mu"], np.abs(hp["sigma"]), hp["df"], hp["a"], hp["b"]
    with np.errstate(invalid="ignore"):
        zero_df = df == 0
        pos_df = (df > 0) & np.isfinite(df)
        box = np.isfinite(a) & np.isfinite(b) & ~np.isfinite(mu) & np.isfinite(sigma)
        fixed = lb == ub
        sb = box & zero_df
        sb_t = box & pos_df
        uni = ~np.isfinite(mu) & ~np.isfinite(sigma)
        gauss = ~uni & ~sb & zero_df & np.isfinite(sigma)
        stud = ~uni & ~sb_t & pos_df
    return dict(fixed=fixed, sb=sb, sb_t=sb_t, uni=uni, gauss=gauss, stud=stud)


_PLANS = {}


def _plan(hp, lb, ub, norm_const):
    """Everything in log_priors that does not depend on ``hyp`` (index sets, which classes are present, the
    elementwise constants of the Gaussian and Student-t terms, the normalisation sum), computed once per
    (priors, bounds): the sampler and the optimiser call log_priors thousands of times per fit with the same
    priors, and the index bookkeeping was a quarter of a fit's wall time at N = 300.  Cached values are the
    same elementwise expressions, so results are bit-identical."""
    arrs = (hp["mu"], hp["sigma"], hp["df"], hp["a"], hp["b"], lb, ub, norm_const)
    key = b"".join(np.ascontiguousarray(x, dtype=float).tobytes() for x in arrs)
    p = _PLANS.get(key)
    if p is None:
        if len(_PLANS) > 64:
            _PLANS.clear()
        sigma, df = np.abs(hp["sigma"]), hp["df"]
        ix = classify(hp, lb, ub)
        g, t = ix["gauss"], ix["stud"]
        p = dict(ix=ix, has={k: bool(np.any(v)) for k, v in ix.items()}, gt=g | t)
        p["has_gt"] = bool(np.any(p["gt"]))
        if p["has"]["gauss"]:
            p["g_log"] = np.log(2 * np.pi * sigma[g] ** 2)
        if p["has"]["stud"]:
            p["t_gam"] = np.sum(sps.gammaln(0.5 * (df[t] + 1)) - sps.gammaln(0.5 * df[t]))
            p["t_c"] = -0.5 * np.log(np.pi * df[t]) - np.log(sigma[t])
        p["log_nc"] = np.sum(np.log(norm_const))
        _PLANS[key] = p
    return p


def log_priors(hyp, hp, lb, ub, norm_const, compute_grad=False):
    """Sum of log prior densities (and gradient) at ``hyp`` -- gaussian_process.py:1275-1466."""
    hyp = np.asarray(hyp, dtype=float)
    mu, sigma, df, a, b = hp["mu"], np.abs(hp["sigma"]), hp["df"], hp["a"], hp["b"]
    plan = _plan(hp, lb, ub, norm_const)
    ix, has = plan["ix"], plan["has"]
    lp = 0
    dlp = np.zeros(hyp.shape) if compute_grad else None

    gt = plan["gt"]
    z2 = np.zeros(hyp.shape)
    if plan["has_gt"]:
        z2[gt] = ((hyp[gt] - mu[gt]) / sigma[gt]) ** 2

    if has["fixed"]:
        if np.any(hyp[ix["fixed"]] != lb[ix["fixed"]]):
            lp = -np.inf
        if compute_grad:
            dlp[ix["fixed"]] = np.nan

    def box_part(sel, student):
        nonlocal lp
        if student:
            C = 1.0 + (b[sel] - a[sel]) * _t_norm(df[sel], sigma[sel])
        else:
            C = 1.0 + (b[sel] - a[sel]) / (sigma[sel] * _S2PI)
        # NOTE: C is the vector over ALL dimensions of this class and is broadcast, as in the reference
        # (:1346-1356, :1391-1413), against the subset of those dimensions that lies outside / inside
        # [a, b].  With several smooth-box dimensions in different regions that either counts terms
        # several times (subset of size 1) or raises numpy's broadcasting ValueError -- the reference's
        # arithmetic, reproduced here so that a seeded fit is identical.
        below = (hyp < a) & sel
        above = (hyp > b) & sel
        inside = (hyp >= a) & (hyp <= b) & sel
        zz = np.zeros(hyp.shape)
        zz[below] = ((hyp[below] - a[below]) / sigma[below]) ** 2
        zz[above] = ((hyp[above] - b[above]) / sigma[above]) ** 2
        out = below | above
        if student:
            for grp in (out, inside):
                if np.any(grp):
                    lp += np.sum(sps.gammaln(0.5 * (df[grp] + 1)) - sps.gammaln(0.5 * df[grp]))
                    tail = -0.5 * (df[grp] + 1) * np.log1p(zz[grp] / df[grp]) if grp is out else 0.0
                    lp += np.sum(-0.5 * np.log(np.pi * df[grp]) - np.log(C * sigma[grp]) + tail)
        else:
            if np.any(out):
                lp -= 0.5 * np.sum(np.log(C**2 * 2 * np.pi * sigma[out] ** 2) + zz[out])
            if np.any(inside):
                lp -= np.sum(np.log(C * sigma[inside]) + np.log(_S2PI))
        if compute_grad:
            for grp, edge in ((below, a), (above, b)):
                if np.any(grp):
                    if student:  # same operation order as the reference: results are compared bit for bit
                        dlp[grp] = (-(df[grp] + 1) / df[grp] / (1 + zz[grp] / df[grp]) * (hyp[grp] - edge[grp])
                                    / sigma[grp] ** 2)
                    else:
                        dlp[grp] = -(hyp[grp] - edge[grp]) / sigma[grp] ** 2

    if has["sb"]:
        box_part(ix["sb"], False)
    if has["sb_t"]:
        box_part(ix["sb_t"], True)

    g = ix["gauss"]
    if has["gauss"]:
        lp -= 0.5 * np.sum(plan["g_log"] + z2[g])
        if compute_grad:
            dlp[g] = -(hyp[g] - mu[g]) / sigma[g] ** 2
    t = ix["stud"]
    if has["stud"]:
        lp += plan["t_gam"]
        lp += np.sum(plan["t_c"] - 0.5 * (df[t] + 1) * np.log1p(z2[t] / df[t]))
        if compute_grad:
            dlp[t] = -(df[t] + 1) / df[t] / (1 + z2[t] / df[t]) * (hyp[t] - mu[t]) / sigma[t] ** 2

    lp -= plan["log_nc"]
    if compute_grad:
        return lp, dlp
    return lp


def log_priors_rows(H, hp, lb, ub, norm_const, compute_grad=False):
    """``log_priors`` for every row of ``H`` (S, hyp_N) at once: lp (S,), dlp (S, hyp_N) | None.

    The design stage of a fit evaluates 1024 rows and the row loop over ``log_priors`` (30 us of NumPy
    bookkeeping per row) was a sixth of a whole fit at N = 300.  Every row's value is produced by the same
    elementwise expressions, in the same order, as ``log_priors`` produces it (sums over the dimensions of a class
    are row sums of a contiguous array: the same pairwise routine), so the results are bit-identical -- checked in
    tests/test_priors_cpu.py.  A smooth-box class with more than one dimension falls back to the row loop: there the
    reference's broadcasting (see ``log_priors``) depends on how many of the dimensions lie outside the box in each row."""
    H = np.atleast_2d(np.asarray(H, dtype=float))
    S = H.shape[0]
    plan = _plan(hp, lb, ub, norm_const)
    ix, has = plan["ix"], plan["has"]
    if (has["sb"] and np.count_nonzero(ix["sb"]) > 1) or (has["sb_t"] and np.count_nonzero(ix["sb_t"]) > 1):
        lp = np.empty(S)
        dlp = np.empty(H.shape) if compute_grad else None
        for r in range(S):
            if compute_grad:
                lp[r], dlp[r] = log_priors(H[r], hp, lb, ub, norm_const, True)
            else:
                lp[r] = log_priors(H[r], hp, lb, ub, norm_const, False)
        return lp, dlp
    mu, sigma, df, a, b = hp["mu"], np.abs(hp["sigma"]), hp["df"], hp["a"], hp["b"]
    lp = np.zeros(S)
    dlp = np.zeros(H.shape) if compute_grad else None
    if has["fixed"]:
        f = ix["fixed"]
        lp[np.any(H[:, f] != lb[f], axis=1)] = -np.inf
        if compute_grad:
            dlp[:, f] = np.nan
    for cls, student in (("sb", False), ("sb_t", True)):
        if not has[cls]:
            continue
        sel = ix[cls]
        d = int(np.flatnonzero(sel)[0])
        h = H[:, d]
        ad, bd, sd = a[sel], b[sel], sigma[sel]  # length-1 arrays: the expressions below are log_priors' own
        below, above = h < ad[0], h > bd[0]
        inside = (h >= ad[0]) & (h <= bd[0])
        zz = np.zeros(S)
        zz[below] = ((h[below] - ad[0]) / sd[0]) ** 2
        zz[above] = ((h[above] - bd[0]) / sd[0]) ** 2
        out = below | above
        if student:
            dfd = df[sel]
            C = 1.0 + (bd - ad) * _t_norm(dfd, sd)
            gam = np.sum(sps.gammaln(0.5 * (dfd + 1)) - sps.gammaln(0.5 * dfd))
            base = -0.5 * np.log(np.pi * dfd) - np.log(C * sd)
            lp[out] += gam
            lp[out] += (base + (-0.5 * (dfd + 1) * np.log1p(zz[out] / dfd)))
            lp[inside] += gam
            lp[inside] += np.sum(base + 0.0)
            if compute_grad:
                for grp, edge in ((below, ad), (above, bd)):
                    dlp[grp, d] = (-(dfd + 1) / dfd / (1 + zz[grp] / dfd) * (h[grp] - edge) / sd ** 2)
        else:
            C = 1.0 + (bd - ad) / (sd * _S2PI)
            lp[out] -= 0.5 * (np.log(C ** 2 * 2 * np.pi * sd ** 2) + zz[out])
            lp[inside] -= np.sum(np.log(C * sd) + np.log(_S2PI))
            if compute_grad:
                for grp, edge in ((below, ad), (above, bd)):
                    dlp[grp, d] = -(h[grp] - edge) / sd ** 2
    if has["gauss"]:
        g = ix["gauss"]
        Hg = H[:, g]
        z2 = ((Hg - mu[g]) / sigma[g]) ** 2
        lp -= 0.5 * np.sum(plan["g_log"] + z2, axis=1)
        if compute_grad:
            dlp[:, g] = -(Hg - mu[g]) / sigma[g] ** 2
    if has["stud"]:
        t = ix["stud"]
        Ht = H[:, t]
        z2 = ((Ht - mu[t]) / sigma[t]) ** 2
        lp += plan["t_gam"]
        lp += np.sum(plan["t_c"] - 0.5 * (df[t] + 1) * np.log1p(z2 / df[t]), axis=1)
        if compute_grad:
            dlp[:, t] = -(df[t] + 1) / df[t] / (1 + z2 / df[t]) * (Ht - mu[t]) / sigma[t] ** 2
    lp -= plan["log_nc"]
    return lp, dlp

"""Space-filling initial design, evaluated in ONE batch on the device.

Reference: gpyreg/f_min_fill.py:11-180.  The reference draws the design (Sobol points
pushed through the inverse cdf of each hyperparameter's prior), then evaluates the
objective at every row in a Python loop (:174-176) -- 1024 independent NLL evaluations
by default, the largest block of core calls inside ``GP.fit``.  Here the design is
generated the same way (same SciPy Sobol sequence, same use of the global NumPy RNG for
the column shuffle, so a seeded run produces the same design) and the objective is
called ONCE with the whole design matrix (``f_batch``), which maps onto
``GP.nll_batch``.  The returned ordering (``argsort`` of the values) is what the caller's
start selection (gaussian_process.py:1109-1125) depends on.
"""

from __future__ import annotations

import warnings

import numpy as np
import scipy.stats as sst

from .priors import (
    _S2PI,
    _t_norm,
    smoothbox_cdf,
    smoothbox_student_t_cdf,
    uuinv,
)


def _piecewise_ppf(q, C, a, b, mid, tail_ppf):
    """priors.smoothbox_ppf / smoothbox_student_t_ppf for a whole column at once (the scalar functions, called
    once per design point, were a third of the design stage): the same three branches, each evaluated with the
    same elementwise expressions on the points that fall into it."""
    q = np.asarray(q, dtype=float)
    out = np.empty(q.shape)
    low = q < 0.5 / C
    mid_m = ~low & (q <= (C - 0.5) / C)
    high = ~low & ~mid_m
    if np.any(low):
        out[low] = tail_ppf(C * q[low], a)
    if np.any(mid_m):
        out[mid_m] = mid(q[mid_m])
    if np.any(high):
        out[high] = tail_ppf(C * q[high] - (C - 1), b)
    return out


def design_points(x0, LB, UB, PLB, PUB, hprior, N, design="sobol"):
    """The N x hyp_N design matrix: the given rows ``x0`` (clipped to the bounds) followed
    by N - len(x0) quasi-random points mapped through each dimension's prior."""
    if design is None:
        design = "sobol"
    N0 = x0.shape[0]
    n_vars = int(np.max([x0.shape[1], np.size(LB), np.size(UB), np.size(PLB), np.size(PUB)]))
    x0 = np.minimum(np.maximum(x0, LB), UB)
    if N <= N0:
        return x0
    n_new = N - N0
    if design == "sobol":
        sampler = sst.qmc.Sobol(d=n_vars, scramble=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            S = sampler.random(n=n_new + 1)[1:, :]  # drop the all-zeros first point
        np.random.shuffle(S.T)  # permute the columns (global RNG, like the reference)
    elif design == "rand":
        S = np.random.uniform(size=(n_new, n_vars))
    else:
        raise ValueError("Unknown design: got " + design + ' and expected either "sobol" or "rand"')

    sX = np.zeros((n_new, n_vars))
    for i in range(n_vars):
        mu, sigma, a, b = hprior["mu"][i], hprior["sigma"][i], hprior["a"][i], hprior["b"][i]
        u = S[:, i]
        if not np.isfinite(mu) and not np.isfinite(sigma):  # no prior: uniform-ish
            if np.isfinite(LB[i]) and np.isfinite(UB[i]):
                if LB[i] == UB[i]:
                    sX[:, i] = LB[i]
                else:
                    sX[:, i] = uuinv(u, [LB[i], PLB[i], PUB[i], UB[i]], 0.5 ** (1 / n_vars))
            else:
                sX[:, i] = u * (PUB[i] - PLB[i]) + PLB[i]
            continue
        df = hprior["df"][i]
        df = 3 if not np.isfinite(df) else np.minimum(df, 3)  # heavy tails for the design
        if np.isfinite(a) and np.isfinite(b):  # smooth-box families, truncated to [LB, UB]
            if df == 0:
                lo, hi = smoothbox_cdf(LB[i], sigma, a, b), smoothbox_cdf(UB[i], sigma, a, b)
                q = lo + (hi - lo) * u
                C = 1.0 + (b - a) / (sigma * _S2PI)
                sX[:, i] = _piecewise_ppf(q, C, a, b, lambda v: (v * C - 0.5) * sigma * _S2PI + a,
                                          lambda v, edge: sst.norm.ppf(v, loc=edge, scale=sigma))
            else:
                lo = smoothbox_student_t_cdf(LB[i], df, sigma, a, b)
                hi = smoothbox_student_t_cdf(UB[i], df, sigma, a, b)
                q = lo + (hi - lo) * u
                c = _t_norm(df, sigma)
                C = 1.0 + (b - a) * c
                sX[:, i] = _piecewise_ppf(q, C, a, b, lambda v: (v * C - 0.5) / c + a,
                                          lambda v, edge: sst.t.ppf(v, df, loc=edge, scale=sigma))
        elif df == 0:  # Gaussian
            lo, hi = sst.norm.cdf((LB[i] - mu) / sigma), sst.norm.cdf((UB[i] - mu) / sigma)
            sX[:, i] = sst.norm.ppf(lo + (hi - lo) * u) * sigma + mu
        else:  # Student-t
            lo, hi = sst.t.cdf((LB[i] - mu) / sigma, df), sst.t.cdf((UB[i] - mu) / sigma, df)
            sX[:, i] = sst.t.ppf(lo + (hi - lo) * u, df) * sigma + mu
    return np.concatenate([x0, sX])


def f_min_fill(f_batch, x0, LB, UB, PLB, PUB, hprior, N, design=None):
    """Design + one batched evaluation + sort (reference f_min_fill.py:11-180).

    ``f_batch(X)`` takes the (N, hyp_N) design and returns the N objective values (the
    reference takes a scalar ``f`` and loops).  Returns X sorted by value, and the values.
    """
    X = design_points(x0, LB, UB, PLB, PUB, hprior, N, design)
    y = np.full((N,), np.inf)
    y[: X.shape[0]] = np.asarray(f_batch(X), dtype=float).ravel()
    order = np.argsort(y)
    return X[order, :], y[order]

"""Mean-function plugins of the GP (the reference's ``gpyreg.mean_functions``: ZeroMean :6-131,
ConstantMean :134-260, NegativeQuadratic :263-397, recommended bounds :400-459).

Boundary plugins of the hot path: O(N*D) host arithmetic whose results -- the
mean vector m and its gradient dm -- are inputs of the device core, exactly where the reference's
``GP.__core_computation`` evaluates them (gaussian_process.py:2379-2387).

Each class is described by a LAYOUT (named hyperparameter blocks) and evaluated for MANY
hyperparameter vectors at once (``values``: every row of a (S, mean_N) array in one NumPy pass --
what ``GP.nll_batch`` needs for a design of 1024 rows); the reference's one-vector ``compute`` is
the S = 1 case of it.  Names, shapes, return conventions (``ZeroMean`` returns ``dm = []``) and
error messages are the reference's, and the arithmetic is ordered so that values are bit-identical
(tests/test_abi_cpu.py checks ``array_equal`` against reference-pinned values).
"""

import numpy as np


class _Mean:
    """Plumbing shared by the mean functions: hyperparameter layout, argument checks, the
    one-vector ``compute`` on top of the batched ``values``, and the recommended-bounds record
    assembled from one row of (LB, UB, PLB, PUB, x0) per hyperparameter block."""

    @classmethod
    def layout(cls, D: int):
        """[(block name, width), ...] in hyperparameter order."""
        raise NotImplementedError

    @classmethod
    def hyperparameter_count(cls, D: int) -> int:
        return sum(width for _, width in cls.layout(D))

    @classmethod
    def hyperparameter_info(cls, D: int):
        return list(cls.layout(D))

    def values(self, hyp_rows: np.ndarray, X: np.ndarray, compute_grad: bool = False):
        """m (S, N) and, with ``compute_grad``, dm (S, N, mean_N) for the rows of ``hyp_rows`` (S, mean_N)."""
        raise NotImplementedError

    def compute(self, hyp: np.ndarray, X: np.ndarray, compute_grad: bool = False):
        """The reference's call: one hyperparameter vector -> m (N,) [, dm (N, mean_N)]."""
        want = self.hyperparameter_count(X.shape[1])
        if hyp.size != want:
            raise ValueError(f"Expected {want} mean function hyperparameters, {hyp.size} passed instead.")
        if hyp.ndim != 1:
            raise ValueError("Mean function output is available only for one-sample hyperparameter inputs.")
        if not compute_grad:
            return self.values(hyp[None, :], X)[0]
        m, dm = self.values(hyp[None, :], X, True)
        return m[0], (dm[0] if want else [])

    def _bound_rows(self, X: np.ndarray, y: np.ndarray):
        """One (LB, UB, PLB, PUB, x0) tuple per block of ``layout``; x0 = None: midpoint of the plausible box."""
        return []

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        if np.size(y) <= 1:  # no observations yet: a unit range
            y = np.array([0, 1])
        keys = ("LB", "UB", "PLB", "PUB", "x0")
        cols = {k: [] for k in keys}
        for (_, width), row in zip(self.layout(X.shape[1]), self._bound_rows(X, y)):
            for k, v in zip(keys, row):
                cols[k].append(np.full((width,), np.nan if v is None else v, dtype=float))
        out = {k: (np.concatenate(v) if v else np.zeros((0,))) for k, v in cols.items()}
        open_x0 = np.isnan(out["x0"])
        out["x0"][open_x0] = 0.5 * (out["PLB"][open_x0] + out["PUB"][open_x0])
        return out


class ZeroMean(_Mean):
    """m(x) = 0: no hyperparameters."""

    @classmethod
    def layout(cls, D):
        return []

    def values(self, hyp_rows, X, compute_grad=False):
        m = np.zeros((hyp_rows.shape[0], X.shape[0]))
        return (m, np.zeros(m.shape + (0,))) if compute_grad else m


class ConstantMean(_Mean):
    """m(x) = m0."""

    @classmethod
    def layout(cls, D):
        return [("mean_const", 1)]

    def values(self, hyp_rows, X, compute_grad=False):
        N = X.shape[0]
        m = hyp_rows[:, 0:1] * np.ones((1, N))
        return (m, np.ones(m.shape + (1,))) if compute_grad else m

    def _bound_rows(self, X, y):
        lo, hi = np.min(y), np.max(y)
        span = hi - lo
        return [(lo - 0.5 * span, hi + 0.5 * span, np.quantile(y, 0.1), np.quantile(y, 0.9), np.median(y))]


class NegativeQuadratic(_Mean):
    """m(x) = m0 - 1/2 sum_d ((x_d - xm_d) / omega_d)^2, hyperparameters [m0 | xm (D) | log omega (D)]."""

    @classmethod
    def layout(cls, D):
        return [("mean_const", 1), ("mean_location", D), ("mean_log_scale", D)]

    def values(self, hyp_rows, X, compute_grad=False):
        D = X.shape[1]
        centre = hyp_rows[:, None, 1:1 + D]                  # (S, 1, D)
        omega = np.exp(hyp_rows[:, None, 1 + D:1 + 2 * D])   # (S, 1, D)
        delta = X[None, :, :] - centre                       # (S, N, D)
        z2 = (delta / omega) ** 2
        m = hyp_rows[:, 0:1] - 0.5 * np.sum(z2, 2)
        if not compute_grad:
            return m
        dm = np.empty(m.shape + (1 + 2 * D,))
        dm[:, :, 0] = 1.0              # d/d m0
        dm[:, :, 1:1 + D] = delta / omega**2   # d/d xm
        dm[:, :, 1 + D:] = z2          # d/d log omega
        return m, dm

    def _bound_rows(self, X, y):
        xlo, xhi = np.min(X), np.max(X)
        ylo, yhi = np.min(y), np.max(y)
        width, span = xhi - xlo, yhi - ylo
        tiny, big = 1e-6, np.exp(3)
        lw = np.log(width)
        return [
            (ylo, yhi + span, np.median(y), yhi, np.quantile(y, 0.9)),
            (xlo - 0.5 * width, xhi + 0.5 * width, xlo, xhi, np.median(X)),
            (lw + np.log(tiny), lw + np.log(big), lw + 0.5 * np.log(tiny), lw, np.log(np.std(X, ddof=1))),
        ]

"""Mean-function plugins (reference: gpyreg/mean_functions.py).

These are the O(N*D) boundary plugins of the hot path: they are
evaluated on the host and their values (m, dm) are handed to the device core, which
is what the reference's GP.__core_computation does with them
(gaussian_process.py:2379-2387).  Same names, shapes and error messages.
"""

import numpy as np


def _check(hyp, mean_N, last_line_joined=False):
    if hyp.size != mean_N:
        raise ValueError(
            f"Expected {mean_N} mean function hyperparameters, "
            f"{hyp.size} passed instead."
        )
    if hyp.ndim != 1:
        raise ValueError(
            "Mean function output is available only for "
            "one-sample hyperparameter inputs."
        )


class ZeroMean:
    """m(x) = 0 (reference :6-131)."""

    def __init__(self):
        pass

    @staticmethod
    def hyperparameter_count(_):
        return 0

    @staticmethod
    def hyperparameter_info(_):
        return []

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        return _bounds_info_helper(self.hyperparameter_count(X.shape[1]), X, y, 0)

    def compute(self, hyp: np.ndarray, X: np.ndarray, compute_grad: bool = False):
        N, D = X.shape
        _check(hyp, self.hyperparameter_count(D))
        m = np.zeros((N,))
        if compute_grad:
            return m, []
        return m


class ConstantMean:
    """m(x) = m0 (reference :134-260)."""

    def __init__(self):
        pass

    @staticmethod
    def hyperparameter_count(_):
        return 1

    @staticmethod
    def hyperparameter_info(_):
        return [("mean_const", 1)]

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        return _bounds_info_helper(self.hyperparameter_count(X.shape[1]), X, y, 1)

    def compute(self, hyp: np.ndarray, X: np.ndarray, compute_grad: bool = False):
        N, D = X.shape
        _check(hyp, self.hyperparameter_count(D))
        m = hyp[0] * np.ones((N,))
        if compute_grad:
            return m, np.ones((N, 1))
        return m


class NegativeQuadratic:
    """m(x) = m0 - 1/2 sum(((x - xm)/omega)^2) (reference :263-397)."""

    def __init__(self):
        pass

    @staticmethod
    def hyperparameter_count(D: int):
        return 1 + 2 * D

    @staticmethod
    def hyperparameter_info(D: int):
        return [("mean_const", 1), ("mean_location", D), ("mean_log_scale", D)]

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        return _bounds_info_helper(self.hyperparameter_count(X.shape[1]), X, y, 2)

    def compute(self, hyp: np.ndarray, X: np.ndarray, compute_grad: bool = False):
        N, D = X.shape
        mean_N = self.hyperparameter_count(D)
        _check(hyp, mean_N)
        m_0 = hyp[0]
        x_m = hyp[1 : 1 + D]
        omega = np.exp(hyp[1 + D : 1 + 2 * D])
        z_2 = ((X - x_m) / omega) ** 2
        m = m_0 - 0.5 * np.sum(z_2, 1)
        if compute_grad:
            dm = np.zeros((N, mean_N))
            dm[:, 0] = np.ones((N,))
            dm[:, 1 : D + 1] = (X - x_m) / omega**2
            dm[:, D + 1 :] = z_2
            return m, dm
        return m


def _bounds_info_helper(mean_N, X, y, idx):
    """Recommended bounds (reference :400-459); idx 0 zero, 1 const, 2 negquad."""
    _, D = X.shape
    tol = 1e-6
    big = np.exp(3)
    LB = np.full((mean_N,), -np.inf)
    UB = np.full((mean_N,), np.inf)
    PLB = np.full((mean_N,), -np.inf)
    PUB = np.full((mean_N,), np.inf)
    x0 = np.full((mean_N,), np.nan)

    w = np.max(X) - np.min(X)
    if np.size(y) <= 1:
        y = np.array([0, 1])
    h = np.max(y) - np.min(y)

    if idx == 1:
        LB[0] = np.min(y) - 0.5 * h
        UB[0] = np.max(y) + 0.5 * h
        PLB[0] = np.quantile(y, 0.1)
        PUB[0] = np.quantile(y, 0.9)
        x0[0] = np.median(y)
    elif idx == 2:
        LB[0] = np.min(y)
        UB[0] = np.max(y) + h
        PLB[0] = np.median(y)
        PUB[0] = np.max(y)
        x0[0] = np.quantile(y, 0.9)
        LB[1 : 1 + D] = np.min(X) - 0.5 * w
        UB[1 : 1 + D] = np.max(X) + 0.5 * w
        PLB[1 : 1 + D] = np.min(X)
        PUB[1 : 1 + D] = np.max(X)
        x0[1 : 1 + D] = np.median(X)
        LB[1 + D : mean_N] = np.log(w) + np.log(tol)
        UB[1 + D : mean_N] = np.log(w) + np.log(big)
        PLB[1 + D : mean_N] = np.log(w) + 0.5 * np.log(tol)
        PUB[1 + D : mean_N] = np.log(w)
        x0[1 + D : mean_N] = np.log(np.std(X, ddof=1))

    i_nan = np.isnan(x0)
    x0[i_nan] = 0.5 * (PLB[i_nan] + PUB[i_nan])
    return {"LB": LB, "PLB": PLB, "UB": UB, "PUB": PUB, "x0": x0}

"""Isotropic (single length-scale) covariance plugins
(reference: gpyreg/isotropic_covariance_functions.py:7-267), computed on the
device like their ARD parents.  Class hierarchy mirrors the reference because
``GP.quad`` tests ``isinstance(cov, SquaredExponential)``."""

import numpy as np

from . import _lib
from .covariance_functions import AbstractKernel, Matern, SquaredExponential, _fill_x0


class AbstractIsotropicKernel(AbstractKernel):
    """Two hyperparameters: log length-scale, log output scale (reference :7-83)."""

    def hyperparameter_count(self, D: int):
        return 2

    def hyperparameter_info(self, D: int):
        return [
            ("covariance_log_lengthscale", 1),
            ("covariance_log_outputscale", 1),
        ]

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        cov_N = self.hyperparameter_count(X.shape[1])
        return _isotropic_bounds_info_helper(cov_N, X, y)


class MaternIsotropic(AbstractIsotropicKernel, Matern):
    """Isotropic Matern kernel (reference :86-161)."""

    _gpc_kernel_id = _lib.K_MATERN_ISO

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        return self._device_compute(hyp, X, X_star, compute_diag, compute_grad)


class SquaredExponentialIsotropic(AbstractIsotropicKernel, SquaredExponential):
    """Isotropic squared exponential kernel (reference :164-221)."""

    _gpc_kernel_id = _lib.K_SE_ISO

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        return self._device_compute(hyp, X, X_star, compute_diag, compute_grad)


def _isotropic_bounds_info_helper(cov_N, X, y):
    """Reference :224-267: one width (mean over dimensions of the data range)."""
    tol = 1e-6
    LB = np.full((cov_N,), -np.inf)
    UB = np.full((cov_N,), np.inf)
    PLB = np.full((cov_N,), -np.inf)
    PUB = np.full((cov_N,), np.inf)
    x0 = np.full((cov_N,), np.nan)

    width = np.mean(np.max(X, axis=0) - np.min(X, axis=0))
    if np.size(y) <= 1:
        y = np.array([0, 1])
    height = np.max(y) - np.min(y)
    n = cov_N - 1

    LB[0:n] = np.log(width) + np.log(tol)
    UB[0:n] = np.log(width * 10)
    PLB[0:n] = np.log(width) + 0.5 * np.log(tol)
    PUB[0:n] = np.log(width)
    x0[0:n] = np.log(np.std(X, ddof=1))

    LB[n] = np.log(height) + np.log(tol)
    UB[n] = np.log(height * 10)
    PLB[n] = np.log(height) + 0.5 * np.log(tol)
    PUB[n] = np.log(height)
    x0[n] = np.log(np.std(y, ddof=1))

    info = {"LB": LB, "UB": UB, "PLB": PLB, "PUB": PUB, "x0": x0}
    _fill_x0(info)
    return info

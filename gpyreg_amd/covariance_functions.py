"""Covariance-function plugins with the reference's interface
(reference: gpyreg/covariance_functions.py:9-367), evaluated by hand-written HIP
kernels through libgpcore.so (``gpc_kernel``; csrc/covfun.h).

Same class names, ``compute`` signature, return shapes and error messages as the
reference, so the reference's tests read unchanged against this module.  There is
no NumPy implementation behind ``compute``: without the HIP library / a GPU it
raises ``RuntimeError``.
"""

from abc import ABC, abstractmethod

import numpy as np

from . import _lib


class AbstractKernel(ABC):
    """Base class (reference covariance_functions.py:9-128)."""

    # built-in kernels are dispatched to the device by id; user subclasses that
    # implement compute() in Python keep _gpc_kernel_id = None
    _gpc_kernel_id = None
    _gpc_degree = 0

    @abstractmethod
    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        """K (N,N) | (N,M) | (N,1) and optionally dK (N,N,cov_N)."""

    def hyperparameter_count(self, D: int):
        return D + 1

    def hyperparameter_info(self, D: int):
        return [
            ("covariance_log_lengthscale", D),
            ("covariance_log_outputscale", 1),
        ]

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        cov_N = self.hyperparameter_count(X.shape[1])
        return _bounds_info_helper(cov_N, X, y)

    # ---- shared argument checks + device dispatch --------------------------------
    def _check_hyp(self, hyp, D):
        cov_N = self.hyperparameter_count(D)
        if hyp.size != cov_N:
            raise ValueError(
                f"Expected {cov_N} covariance function hyperparameters, "
                f"{hyp.size} passed instead."
            )
        if hyp.ndim != 1:
            raise ValueError(
                "Covariance function output is available only for "
                "one-sample hyperparameter inputs."
            )
        return cov_N

    def _device_compute(self, hyp, X, X_star, compute_diag, compute_grad):
        hyp = np.asarray(hyp)
        X = np.asarray(X)
        N, D = X.shape
        self._check_hyp(hyp, D)
        if compute_grad and X_star is not None:
            raise ValueError("X_star should be None when compute_grad is True.")
        ctx = _lib.context()
        if X_star is None and compute_diag:
            K = ctx.kernel(self._gpc_kernel_id, self._gpc_degree, hyp, X, diag=True)
            if compute_grad:
                # reference quirk: zero "distance" (N,1) broadcast against (N,N) planes
                K2, dK = ctx.kernel(self._gpc_kernel_id, self._gpc_degree, hyp, X, grad=True)
                return K, dK
            return K
        if compute_grad:
            return ctx.kernel(self._gpc_kernel_id, self._gpc_degree, hyp, X, grad=True)
        return ctx.kernel(self._gpc_kernel_id, self._gpc_degree, hyp, X, X_star=X_star)


class SquaredExponential(AbstractKernel):
    """Squared exponential ARD kernel (reference :131-186)."""

    _gpc_kernel_id = _lib.K_SE

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        return self._device_compute(hyp, X, X_star, compute_diag, compute_grad)


class Matern(AbstractKernel):
    """Matern ARD kernel of degree 1, 3 or 5 (reference :189-285)."""

    _gpc_kernel_id = _lib.K_MATERN

    def __init__(self, degree: int):
        if degree not in (1, 3, 5):
            raise ValueError(
                "Only degrees 1, 3 and 5 are supported for the "
                "Matern covariance function."
            )
        self.degree = degree
        self._gpc_degree = degree

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        return self._device_compute(hyp, X, X_star, compute_diag, compute_grad)


class RationalQuadraticARD(AbstractKernel):
    """Rational quadratic ARD kernel (reference :288-421)."""

    _gpc_kernel_id = _lib.K_RQ

    def hyperparameter_count(self, D: int):
        return D + 2

    def hyperparameter_info(self, D: int):
        return [
            ("covariance_log_lengthscale", D),
            ("covariance_log_outputscale", 1),
            ("covariance_log_shape", 1),
        ]

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        return self._device_compute(hyp, X, X_star, compute_diag, compute_grad)

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        # same recipe as the helper, plus the shape parameter initialised as in BADS
        # (reference :369-421, including its use of index D for the plausible upper
        # bound of the shape entry)
        cov_N = self.hyperparameter_count(X.shape[1])
        D = X.shape[1]
        info = _bounds_info_helper(cov_N, X, y, fill_x0=False)
        info["LB"][-1] = -5.0
        info["UB"][-1] = 5
        info["PLB"][-1] = -5.0
        info["PUB"][D] = 5.0
        info["x0"][-1] = 1.0
        _fill_x0(info)
        return info


def _fill_x0(info):
    x0 = info["x0"]
    i_nan = np.isnan(x0)
    x0[i_nan] = 0.5 * (info["PLB"][i_nan] + info["PUB"][i_nan])


def _bounds_info_helper(cov_N, X, y, fill_x0=True):
    """Recommended bounds for [log lengthscales (D), log output scale]
    (reference :424-463): scales span [tol, 10] x the data width/height."""
    _, D = X.shape
    tol = 1e-6
    LB = np.full((cov_N,), -np.inf)
    UB = np.full((cov_N,), np.inf)
    PLB = np.full((cov_N,), -np.inf)
    PUB = np.full((cov_N,), np.inf)
    x0 = np.full((cov_N,), np.nan)

    width = np.max(X, axis=0) - np.min(X, axis=0)
    if np.size(y) <= 1:
        y = np.array([0, 1])
    height = np.max(y) - np.min(y)

    LB[0:D] = np.log(width) + np.log(tol)
    UB[0:D] = np.log(width * 10)
    PLB[0:D] = np.log(width) + 0.5 * np.log(tol)
    PUB[0:D] = np.log(width)
    x0[0:D] = np.log(np.std(X, ddof=1))

    LB[D] = np.log(height) + np.log(tol)
    UB[D] = np.log(height * 10)
    PLB[D] = np.log(height) + 0.5 * np.log(tol)
    PUB[D] = np.log(height)
    x0[D] = np.log(np.std(y, ddof=1))

    info = {"LB": LB, "UB": UB, "PLB": PLB, "PUB": PUB, "x0": x0}
    if fill_x0:
        _fill_x0(info)
    return info

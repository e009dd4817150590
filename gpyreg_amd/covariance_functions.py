"""Covariance-function plugins with the reference's interface
(reference: gpyreg/covariance_functions.py:9-367), evaluated by hand-written HIP
kernels through libgpcore.so (``gpc_kernel``; csrc/covfun.h).

Same class names, ``compute`` signature, return shapes and error messages as the
reference, so the reference's tests read unchanged against this module.  There is
no NumPy implementation behind ``compute``: without the HIP library / a GPU it
raises ``RuntimeError``.

How the module is put together: a kernel class DESCRIBES its hyperparameter vector as a ``layout`` -- blocks of
(name, width as a function of D, which data extent the block is a log-scale of) -- and everything the GP asks of a
plugin besides ``compute`` (count, info, recommended bounds) is derived from that description by the base class.
"""

from abc import ABC, abstractmethod

import numpy as np

from . import _lib

_PER_DIM = "per input dimension"  # block width D; any other width is a constant
_TINY = 1e-6                      # a length or output scale is searched down to this fraction of the data's extent


def _log_scale_box(extent):
    """Search box of a log-scale hyperparameter whose natural unit is ``extent`` (a data range; scalar or per
    dimension): hard bounds [1e-6, 10] x extent, plausible bounds [1e-3, 1] x extent (reference :437-452, same
    floating-point operations: the small factor enters as a sum of logarithms, the 10 as a product)."""
    le = np.log(extent)
    return le + np.log(_TINY), np.log(extent * 10), le + 0.5 * np.log(_TINY), le


def _plugin_input_error(what, wanted, got):
    """The two complaints every plugin makes about its hyperparameter vector (messages asserted by the reference's
    tests, testing/test_covariance_functions.py): wrong length; not one-dimensional."""
    if got.size != wanted:
        raise ValueError(f"Expected {wanted} {what} function hyperparameters, {got.size} passed instead.")
    if got.ndim != 1:
        raise ValueError(f"{what.capitalize()} function output is available only for one-sample hyperparameter inputs.")


class AbstractKernel(ABC):
    """Base class (reference covariance_functions.py:9-128).  A subclass sets ``layout`` (and, for the built-in
    kernels, the device ids); a user-defined kernel written against the reference's protocol overrides the methods
    themselves and keeps ``_gpc_kernel_id = None`` (its own ``compute`` then feeds the device factorization)."""

    _gpc_kernel_id = None
    _gpc_degree = 0
    # (name, width, extent): extent "x" = the inputs' range (per dimension, or averaged for one shared scale),
    # "y" = the observations' range, None = no data-derived box (the class fills it in itself)
    layout = (("covariance_log_lengthscale", _PER_DIM, "x"), ("covariance_log_outputscale", 1, "y"))

    @abstractmethod
    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        """K (N,N) | (N,M) | (N,1) and optionally dK (N,N,cov_N)."""

    def _blocks(self, D):
        return [(name, D if width == _PER_DIM else width, extent) for name, width, extent in self.layout]

    def hyperparameter_count(self, D: int):
        return sum(width for _, width, _ in self._blocks(D))

    def hyperparameter_info(self, D: int):
        return [(name, width) for name, width, _ in self._blocks(D)]

    def _shared_input_range(self, spread):
        return spread  # ARD: one range per dimension

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        """LB / UB / PLB / PUB / x0 over the hyperparameter vector (reference :96-128, :424-463)."""
        D = X.shape[1]
        n = self.hyperparameter_count(D)
        box = {"LB": np.full(n, -np.inf), "UB": np.full(n, np.inf), "PLB": np.full(n, -np.inf),
               "PUB": np.full(n, np.inf), "x0": np.full(n, np.nan)}
        x_range = self._shared_input_range(np.max(X, axis=0) - np.min(X, axis=0))
        obs = y if np.size(y) > 1 else np.array([0, 1])  # (no or one observation: a unit output range)
        y_range = np.max(obs) - np.min(obs)
        at = 0
        for _, width, extent in self._blocks(D):
            sl = slice(at, at + width)
            at += width
            if extent is None:
                continue
            lo, hi, plo, phi = _log_scale_box(x_range if extent == "x" else y_range)
            box["LB"][sl], box["UB"][sl], box["PLB"][sl], box["PUB"][sl] = lo, hi, plo, phi
            # start: the spread of the data itself (of ALL inputs together for the length scales, as the reference does)
            box["x0"][sl] = np.log(np.std(X if extent == "x" else obs, ddof=1))
        self._finish_bounds(box, D)
        unset = np.isnan(box["x0"])
        box["x0"][unset] = 0.5 * (box["PLB"][unset] + box["PUB"][unset])
        return box

    def _finish_bounds(self, box, D):
        """Entries the data do not speak about (hook for subclasses)."""

    # ---- device dispatch ----------------------------------------------------------
    def _device_compute(self, hyp, X, X_star, compute_diag, compute_grad):
        hyp = np.asarray(hyp)
        X = np.asarray(X)
        _plugin_input_error("covariance", self.hyperparameter_count(X.shape[1]), hyp)
        if compute_grad and X_star is not None:
            raise ValueError("X_star should be None when compute_grad is True.")
        run = _lib.context().kernel
        kid, deg = self._gpc_kernel_id, self._gpc_degree
        if compute_grad:
            full = run(kid, deg, hyp, X, grad=True)
            if X_star is None and compute_diag:
                # the reference's diagonal branch broadcasts its (N,1) "distance" against (N,N) planes: the gradient
                # it returns there is the full-matrix one
                return run(kid, deg, hyp, X, diag=True), full[1]
            return full
        if X_star is None and compute_diag:
            return run(kid, deg, hyp, X, diag=True)
        return run(kid, deg, hyp, X, X_star=X_star)


class _BuiltIn(AbstractKernel):
    """A kernel the library has device code for: ``compute`` is the dispatch above."""

    def compute(self, hyp, X, X_star=None, compute_diag=False, compute_grad=False):
        return self._device_compute(hyp, X, X_star, compute_diag, compute_grad)


class SquaredExponential(_BuiltIn):
    """Squared exponential ARD kernel (reference :131-186)."""

    _gpc_kernel_id = _lib.K_SE


class Matern(_BuiltIn):
    """Matern ARD kernel of degree 1, 3 or 5 (reference :189-285)."""

    _gpc_kernel_id = _lib.K_MATERN
    _DEGREES = (1, 3, 5)

    def __init__(self, degree: int):
        if degree not in self._DEGREES:
            raise ValueError("Only degrees 1, 3 and 5 are supported for the Matern covariance function.")
        self.degree = self._gpc_degree = degree


class RationalQuadraticARD(_BuiltIn):
    """Rational quadratic ARD kernel (reference :288-421)."""

    _gpc_kernel_id = _lib.K_RQ
    layout = AbstractKernel.layout + (("covariance_log_shape", 1, None),)

    def _finish_bounds(self, box, D):
        # the shape parameter: [-5, 5], started at 1 as in BADS (reference :369-421) -- including the reference's slip
        # of writing the plausible UPPER bound of the shape entry to index D, the output scale's slot
        box["LB"][-1], box["UB"][-1], box["PLB"][-1] = -5.0, 5, -5.0
        box["PUB"][D] = 5.0
        box["x0"][-1] = 1.0

"""Isotropic (single length-scale) covariance plugins
(reference: gpyreg/isotropic_covariance_functions.py:7-267), computed on the
device like their ARD parents.  Class hierarchy mirrors the reference because
``GP.quad`` tests ``isinstance(cov, SquaredExponential)``.

An isotropic kernel is its ARD parent with another ``layout``: ONE length scale, whose natural unit is the inputs'
range averaged over the dimensions (reference :233); counts, info and bounds follow from that (covariance_functions.py).
"""

import numpy as np

from . import _lib
from .covariance_functions import AbstractKernel, Matern, SquaredExponential


class AbstractIsotropicKernel(AbstractKernel):
    """Two hyperparameters: log length-scale, log output scale (reference :7-83)."""

    layout = (("covariance_log_lengthscale", 1, "x"), ("covariance_log_outputscale", 1, "y"))

    def _shared_input_range(self, spread):
        return np.mean(spread)


class MaternIsotropic(AbstractIsotropicKernel, Matern):
    """Isotropic Matern kernel (reference :86-161)."""

    _gpc_kernel_id = _lib.K_MATERN_ISO


class SquaredExponentialIsotropic(AbstractIsotropicKernel, SquaredExponential):
    """Isotropic squared exponential kernel (reference :164-221)."""

    _gpc_kernel_id = _lib.K_SE_ISO

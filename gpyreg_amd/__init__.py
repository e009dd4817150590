"""gpyreg_amd -- MI355X-native dense Gaussian-process core behind the gpyreg plugin API.

Drop-in for the dense hot path of acerbilab/gpyreg: ``GP`` with the
``covariance_functions`` / ``isotropic_covariance_functions`` / ``mean_functions`` /
``noise_functions`` plugin modules (same class names as the reference,
gpyreg/__init__.py:3-9).  All O(N^2)/O(N^3) work runs in hand-written HIP kernels for
gfx950 behind the C ABI of include/gpcore.h; there is no CPU fallback.
"""

from . import (
    covariance_functions,
    isotropic_covariance_functions,
    mean_functions,
    noise_functions,
)
from .gaussian_process import GP, Posterior

__all__ = [
    "GP",
    "Posterior",
    "covariance_functions",
    "isotropic_covariance_functions",
    "mean_functions",
    "noise_functions",
]

"""Noise-function plugin (reference: gpyreg/noise_functions.py).

An O(N) boundary plugin of the hot path: evaluated on the host;
its value sn2 (scalar or per-point) and gradient dsn2 are inputs of the device core
(gaussian_process.py:2372-2378).  The scalar-vs-array distinction of the return
value is kept because it selects the branch at gaussian_process.py:2407/:2491.
"""

import numpy as np


class GaussianNoise:
    """Gaussian observation noise: sum of the enabled variance terms
    (reference :6-41).  ``parameters`` = [constant, user-provided (1) / scaled (2),
    rectified-linear output dependent]."""

    def __init__(
        self,
        constant_add: bool = False,
        user_provided_add: bool = False,
        scale_user_provided: bool = False,
        rectified_linear_output_dependent_add: bool = False,
    ):
        self.parameters = np.zeros((3,))
        if constant_add:
            self.parameters[0] = 1
        if user_provided_add:
            self.parameters[1] = 1
            if scale_user_provided:
                self.parameters[1] += 1
        if rectified_linear_output_dependent_add:
            self.parameters[2] = 1

    def hyperparameter_count(self):
        p = self.parameters
        return int(p[0] == 1) + int(p[1] == 2) + 2 * int(p[2] == 1)

    def hyperparameter_info(self):
        info = []
        if self.parameters[0] == 1:
            info.append(("noise_log_scale", 1))
        if self.parameters[1] == 2:
            info.append(("noise_provided_log_multiplier", 1))
        if self.parameters[2] == 1:
            info.append(("noise_rectified_log_multiplier", 2))
        return info

    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        """Recommended bounds (reference :82-177)."""
        _, D = X.shape
        noise_N = self.hyperparameter_count()
        tol = 1e-6
        LB = np.full((noise_N,), -np.inf)
        UB = np.full((noise_N,), np.inf)
        PLB = np.full((noise_N,), -np.inf)
        PUB = np.full((noise_N,), np.inf)
        x0 = np.full((noise_N,), np.nan)
        if np.size(y) <= 1:
            y = np.array([0, 1])
        height = np.max(y) - np.min(y)

        i = 0
        if self.parameters[0] == 1:
            LB[i], UB[i] = np.log(tol), np.log(height)
            PLB[i], PUB[i] = 0.5 * np.log(tol), np.log(np.std(y, ddof=1))
            x0[i] = np.log(1e-3)
            i += 1
        if self.parameters[1] == 2:
            LB[i], UB[i] = np.log(1e-3), np.log(1e3)
            PLB[i], PUB[i] = np.log(0.5), np.log(2)
            x0[i] = np.log(1)
            i += 1
        if self.parameters[2] == 1:
            min_y, max_y = np.min(y), np.max(y)
            LB[i], UB[i] = min_y, max_y
            PLB[i], PUB[i] = min_y, np.maximum(max_y - 5 * D, min_y)
            x0[i] = np.maximum(max_y - 10 * D, min_y)
            i += 1
            LB[i], UB[i] = np.log(1e-3), np.log(0.1)
            PLB[i], PUB[i] = np.log(0.01), np.log(0.1)
            x0[i] = np.log(0.1)
            i += 1

        i_nan = np.isnan(x0)
        x0[i_nan] = 0.5 * (PLB[i_nan] + PUB[i_nan])
        return {"LB": LB, "PLB": PLB, "PUB": PUB, "UB": UB, "x0": x0}

    def compute(self, hyp, X, y, s2=None, compute_grad: bool = False):
        """sn2 (scalar when there is no per-point term, else (N,1)) and dsn2
        ((1|N), noise_N) -- reference :179-283."""
        N, _ = X.shape
        noise_N = self.hyperparameter_count()
        if hyp.size != noise_N:
            raise ValueError(
                f"Expected {noise_N} noise function hyperparameters, "
                f"{hyp.size} passed instead."
            )
        if hyp.ndim != 1:
            raise ValueError(
                "Noise function output is available only for "
                "one-sample hyperparameter inputs."
            )
        p = self.parameters
        dsn2 = None
        if compute_grad:
            rows = N if any(x > 0 for x in p[1:]) else 1
            dsn2 = np.zeros((rows, noise_N))

        i = 0
        if p[0] == 0:
            sn2 = np.spacing(1.0)
        else:
            sn2 = np.exp(2 * hyp[i])
            if compute_grad:
                dsn2[:, i] = 2 * sn2
            i += 1

        if s2 is None:
            s2 = 0
        if p[1] == 1:
            sn2 = sn2 + s2
        elif p[1] == 2:
            sn2 = sn2 + np.exp(hyp[i]) * s2
            if compute_grad:
                dsn2[:, i : i + 1] = np.exp(hyp[i]) * s2
            i += 1

        if p[2] == 1:
            if y is not None:
                y_tresh = hyp[i]
                w2 = np.exp(2 * hyp[i + 1])
                zz = np.maximum(0, y_tresh - y)
                sn2 = sn2 + w2 * zz**2
                if compute_grad:
                    dsn2[:, i : i + 1] = 2 * w2 * (y_tresh - y) * (zz > 0)
                    dsn2[:, i + 1 : i + 2] = 2 * w2 * zz**2
            i += 2

        if compute_grad:
            return sn2, dsn2
        return sn2

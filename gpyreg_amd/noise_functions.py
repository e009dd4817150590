"""Observation-noise plugin of the GP (the reference's ``gpyreg.noise_functions.GaussianNoise``:
constructor :6-41, layout :43-80, recommended bounds :82-177, ``compute`` :179-283).

A boundary plugin of the hot path: O(N) host arithmetic whose results -- the
noise variance sn2 and its gradient dsn2 -- are inputs of the device core
(gaussian_process.py:2372-2378).  The variance is a sum of up to three TERMS, each with its own
hyperparameters:

    constant          exp(2 h)                          1 hyperparameter   (absent: eps, no hyperparameter)
    provided          s2   |   exp(h) * s2              0 | 1              (the caller's per-point estimates)
    rectified-linear  exp(2 h_w) * max(0, h_t - y)^2    2                  (output dependent)

``values`` evaluates all rows of a (S, noise_N) hyperparameter array in one NumPy pass (what
``GP.nll_batch`` needs); the reference's one-vector ``compute`` is its S = 1 case and keeps the
reference's return convention, which the core branches on (gaussian_process.py:2407, :2491): a
SCALAR when no per-point term contributes, an (N, 1) array otherwise; dsn2 has one row or N rows
accordingly.  Arithmetic is ordered so that values are bit-identical to the reference's
(tests/test_abi_cpu.py, ``array_equal`` against reference-pinned values).
"""

import numpy as np


class GaussianNoise:
    def __init__(self, constant_add: bool = False, user_provided_add: bool = False,
                 scale_user_provided: bool = False, rectified_linear_output_dependent_add: bool = False):
        # the reference's encoding, read by GP.__str__ and by callers: [constant, provided (2 = scaled), rectified]
        provided = (2 if scale_user_provided else 1) if user_provided_add else 0
        self.parameters = np.array([1.0 if constant_add else 0.0, float(provided),
                                    1.0 if rectified_linear_output_dependent_add else 0.0])

    # ---- layout ---------------------------------------------------------------------------
    def _terms(self):
        """[(block name, width), ...] of the enabled terms that own hyperparameters."""
        const, provided, rect = self.parameters
        terms = []
        if const == 1:
            terms.append(("noise_log_scale", 1))
        if provided == 2:
            terms.append(("noise_provided_log_multiplier", 1))
        if rect == 1:
            terms.append(("noise_rectified_log_multiplier", 2))
        return terms

    def hyperparameter_count(self) -> int:
        # (= sum of the widths of _terms(); spelled out because it sits on the path of every single evaluation)
        const, provided, rect = self.parameters.tolist()
        return (1 if const == 1 else 0) + (1 if provided == 2 else 0) + (2 if rect == 1 else 0)

    def hyperparameter_info(self):
        return self._terms()

    # ---- recommended bounds ---------------------------------------------------------------
    def get_bounds_info(self, X: np.ndarray, y: np.ndarray):
        D = X.shape[1]
        if np.size(y) <= 1:  # no observations yet: a unit range
            y = np.array([0, 1])
        ylo, yhi = np.min(y), np.max(y)
        tiny = 1e-6
        # one (LB, UB, PLB, PUB, x0) row per hyperparameter, by block; a block's rows are only evaluated when the block
        # is enabled (log(yhi - ylo) and the sample deviation warn on constant or single observations, and the
        # reference computes them for the constant term only)
        table = {
            "noise_log_scale": lambda: [
                (np.log(tiny), np.log(yhi - ylo), 0.5 * np.log(tiny), np.log(np.std(y, ddof=1)), np.log(1e-3))],
            "noise_provided_log_multiplier": lambda: [
                (np.log(1e-3), np.log(1e3), np.log(0.5), np.log(2), np.log(1))],
            "noise_rectified_log_multiplier": lambda: [
                (ylo, yhi, ylo, np.maximum(yhi - 5 * D, ylo), np.maximum(yhi - 10 * D, ylo)),  # threshold
                (np.log(1e-3), np.log(0.1), np.log(0.01), np.log(0.1), np.log(0.1))],          # log slope
        }
        rows = [r for name, _ in self._terms() for r in table[name]()]
        cols = np.array(rows, dtype=float).reshape(len(rows), 5)
        out = {k: cols[:, j].copy() for j, k in enumerate(("LB", "UB", "PLB", "PUB", "x0"))}
        open_x0 = np.isnan(out["x0"])
        out["x0"][open_x0] = 0.5 * (out["PLB"][open_x0] + out["PUB"][open_x0])
        return out

    # ---- values ---------------------------------------------------------------------------
    def per_point(self, y, s2) -> bool:
        """True when sn2 comes out as one value per observation (a provided or an output-dependent term
        is enabled AND its data are there), False when it is a single number."""
        _, provided, rect = self.parameters
        return (provided >= 1 and s2 is not None and not np.isscalar(s2)) or (rect == 1 and y is not None)

    def values(self, hyp_rows: np.ndarray, X: np.ndarray, y, s2=None, compute_grad: bool = False):
        """sn2 for every row of ``hyp_rows`` (S, noise_N): (S, N) when ``per_point(y, s2)`` else (S, 1);
        with ``compute_grad`` also dsn2 (S, R, noise_N), R = N when any per-point term is ENABLED
        (reference :243-246: even if its data are absent), else 1."""
        S, N = hyp_rows.shape[0], X.shape[0]
        const, provided, rect = self.parameters
        noise_N = self.hyperparameter_count()
        wide = self.per_point(y, s2)
        grad = np.zeros((S, N if (provided > 0 or rect > 0) else 1, noise_N)) if compute_grad else None
        col = 0
        if const == 1:
            sn2 = np.exp(2 * hyp_rows[:, col:col + 1])  # (S, 1)
            if compute_grad:
                grad[:, :, col] = 2 * sn2
            col += 1
        else:
            sn2 = np.full((S, 1), np.spacing(1.0))
        given = 0 if s2 is None else (s2 if np.isscalar(s2) else np.reshape(s2, (1, N)))
        if provided == 1:
            sn2 = sn2 + given
        elif provided == 2:
            scaled = np.exp(hyp_rows[:, col:col + 1]) * given
            sn2 = sn2 + scaled
            if compute_grad:
                grad[:, :, col] = scaled
            col += 1
        if rect == 1:
            if y is not None:
                threshold = hyp_rows[:, col:col + 1]
                w2 = np.exp(2 * hyp_rows[:, col + 1:col + 2])
                gap = threshold - np.reshape(y, (1, N))
                below = np.maximum(0, gap)
                sn2 = sn2 + w2 * below**2
                if compute_grad:
                    grad[:, :, col] = 2 * w2 * gap * (below > 0)
                    grad[:, :, col + 1] = 2 * w2 * below**2
            col += 2
        assert sn2.shape == (S, N if wide else 1)
        return (sn2, grad) if compute_grad else sn2

    def compute(self, hyp: np.ndarray, X: np.ndarray, y, s2=None, compute_grad: bool = False):
        """The reference's call: one hyperparameter vector -> sn2 (scalar | (N, 1)) [, dsn2 ((1 | N), noise_N)]."""
        want = self.hyperparameter_count()
        if hyp.size != want:
            raise ValueError(f"Expected {want} noise function hyperparameters, {hyp.size} passed instead.")
        if hyp.ndim != 1:
            raise ValueError("Noise function output is available only for one-sample hyperparameter inputs.")
        out = self.values(hyp[None, :], X, y, s2, compute_grad)
        sn2 = out[0] if compute_grad else out
        sn2 = sn2[0].reshape(-1, 1) if self.per_point(y, s2) else sn2[0, 0]
        return (sn2, out[1][0]) if compute_grad else sn2

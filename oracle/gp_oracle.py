"""CPU oracle for the dense GP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, in plain NumPy/SciPy, the algorithm of the reference
(acerbilab/gpyreg) for the one path this repository accelerates.  It exists only
so that ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg can check (and time) the HIP implementation against it.
Nothing under ``gpyreg_amd/`` may import it; the product path fails loudly when
the HIP library is missing.

Parity pin: every function below is checked bit-for-bit / to 1e-12 against
golden vectors produced by importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_golden.py``).

Reference lines followed (paths relative to the reference checkout):
  covariance_functions.py:135-186   SquaredExponential.compute
  covariance_functions.py:221-285   Matern.compute (degrees 1/3/5, :210-218)
  covariance_functions.py:301-367   RationalQuadraticARD.compute
  isotropic_covariance_functions.py:104-161, :173-221   isotropic variants
  noise_functions.py:179-283        GaussianNoise.compute
  mean_functions.py:82-131, :210-260, :340-397   Zero/Constant/NegativeQuadratic
  gaussian_process.py:2357-2521     GP.__core_computation
  gaussian_process.py:1663-1816     GP.predict
  gaussian_process.py:870-884       GP.update full-recompute loop
  gaussian_process.py:750-844       GP.update rank-one path (one new point)
  core_streamed: the same core with the gradient planes of covariance_functions.py:177-184, :267-283,
                 :349-363 (isotropic :149-158, :216-218) formed ONE AT A TIME instead of as an (N, N, cov_N)
                 tensor -- the only way to a gradient at cfg4 (N = 16384, 22 planes = 47 GB in the reference)

Third-party arithmetic the reference delegates to (source not in the reference
tree): scipy.spatial.distance.{pdist,cdist,squareform}, scipy.linalg.{cholesky,
solve_triangular}; the oracle calls the very same SciPy entry points so that its
floating-point results are those of the reference on the same machine.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg as sla
from scipy.spatial.distance import cdist, pdist, squareform

# --------------------------------------------------------------------------
# model descriptors (plain tuples/dicts; no classes so nothing here can be
# mistaken for the plugin API of the product)
# --------------------------------------------------------------------------

KERNELS = ("se", "matern", "rq", "se_iso", "matern_iso")
MEANS = ("zero", "const", "negquad")


def cov_count(kernel, D: int) -> int:
    """covariance_functions.py:59-73, :291-292; isotropic_...py:14-28.  ``kernel`` may also be an
    object with the reference's covariance protocol (hyperparameter_count / compute): the
    reference calls whatever object it was given (gaussian_process.py:2388-2390)."""
    if hasattr(kernel, "hyperparameter_count"):
        return kernel.hyperparameter_count(D)
    if kernel in ("se", "matern"):
        return D + 1
    if kernel == "rq":
        return D + 2
    if kernel in ("se_iso", "matern_iso"):
        return 2
    raise ValueError(kernel)


def mean_count(mean: str, D: int) -> int:
    """mean_functions.py:12-26, :139-154, :268-283."""
    return {"zero": 0, "const": 1, "negquad": 1 + 2 * D}[mean]


def noise_count(params) -> int:
    """noise_functions.py:43-59.  params = (const, user, rectlin) flags 0/1/2."""
    n = 0
    if params[0] == 1:
        n += 1
    if params[1] == 2:
        n += 1
    if params[2] == 1:
        n += 2
    return n


def _matern_f_df(degree: int):
    """covariance_functions.py:210-218."""
    if degree == 1:
        return (lambda t: 1), (lambda t: 1 / t)
    if degree == 3:
        return (lambda t: 1 + t), (lambda t: 1)
    if degree == 5:
        return (lambda t: 1 + t * (1 + t / 3)), (lambda t: (1 + t) / 3)
    raise ValueError("Only degrees 1, 3 and 5 are supported")


# --------------------------------------------------------------------------
# covariance
# --------------------------------------------------------------------------


def covariance(
    kernel: str,
    hyp: np.ndarray,
    X: np.ndarray,
    X_star: np.ndarray | None = None,
    compute_diag: bool = False,
    compute_grad: bool = False,
    degree: int = 0,
):
    """K (and dK[N,N,cov_N]) exactly as the reference's ``compute`` methods.

    Order of floating-point operations follows the reference line by line
    (scale X first, then SciPy distance, then the kernel function) because the
    golden test is bit-exact.
    """
    if hasattr(kernel, "compute"):  # a covariance OBJECT: call it like the reference does
        return kernel.compute(hyp, X, X_star, compute_diag=compute_diag, compute_grad=compute_grad)
    N, D = X.shape
    cov_N = cov_count(kernel, D)
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
    if compute_grad and X_star is not None:
        raise ValueError("X_star should be None when compute_grad is True.")

    iso = kernel.endswith("_iso")
    if iso:
        ell = np.exp(hyp[0])  # isotropic_...py:127, :196
        sf2 = np.exp(2 * hyp[1])
    else:
        ell = np.exp(hyp[0:D])  # covariance_functions.py:158, :244, :324
        sf2 = np.exp(2 * hyp[D])

    if kernel in ("se", "se_iso"):
        # covariance_functions.py:161-169 / isotropic :199-207
        if X_star is None:
            if compute_diag:
                tmp = np.zeros((N, 1))
            else:
                tmp = squareform(pdist(X / ell, "sqeuclidean"))
        else:
            tmp = cdist(X / ell, X_star / ell, "sqeuclidean")
        K = sf2 * np.exp(-tmp / 2)
        if not compute_grad:
            return K
        dK = np.zeros((cov_N, N, N))
        if iso:
            dK[0] = K * squareform(pdist(X / ell, "sqeuclidean"))  # :216
            dK[1] = 2 * K
        else:
            for i in range(D):  # :177-181
                dK[i] = K * squareform(
                    pdist(np.reshape(X[:, i] / ell[i], (-1, 1)), "sqeuclidean")
                )
            dK[D] = 2 * K
        return K, dK.transpose(1, 2, 0)

    if kernel in ("matern", "matern_iso"):
        f, df = _matern_f_df(degree)
        if iso:
            # isotropic :130-138
            if X_star is None:
                if compute_diag:
                    tmp = np.zeros((N, 1))
                else:
                    tmp = squareform(pdist(X * np.sqrt(degree) / ell))
            else:
                tmp = cdist(
                    X * np.sqrt(degree) / ell, X_star * np.sqrt(degree) / ell
                )
        else:
            # covariance_functions.py:247-257
            S = np.diag(np.sqrt(degree) / ell)
            if X_star is None:
                if compute_diag:
                    tmp = np.zeros((N, 1))
                else:
                    tmp = squareform(pdist(X @ S))
            else:
                tmp = cdist(X @ S, X_star @ S)
        K = sf2 * f(tmp) * np.exp(-tmp)
        if not compute_grad:
            return K
        dK = np.zeros((cov_N, N, N))
        with np.errstate(all="ignore"):
            if iso:
                K_ls = squareform(
                    pdist(np.sqrt(degree) / ell * X, "sqeuclidean")
                )  # :149-151
                dK[0] = sf2 * (df(tmp) * np.exp(-tmp)) * K_ls
                dK[1] = 2 * K
            else:
                for i in range(D):  # :267-280
                    Ki = squareform(
                        pdist(
                            np.reshape(np.sqrt(degree) / ell[i] * X[:, i], (-1, 1)),
                            "sqeuclidean",
                        )
                    )
                    dK[i] = sf2 * (df(tmp) * np.exp(-tmp)) * Ki
                dK[D] = 2 * K
        return K, dK.transpose(1, 2, 0)

    if kernel == "rq":
        alpha = np.exp(hyp[D + 1])  # :326
        S = np.diag(1.0 / ell)
        if X_star is None:
            if compute_diag:
                tmp = np.zeros((N, 1))
            else:
                tmp = squareform(pdist(X @ S, "sqeuclidean"))
        else:
            tmp = cdist(X @ S, X_star @ S, "sqeuclidean")
        M = 1 + 0.5 * tmp / alpha  # :338
        K = sf2 * M ** (-alpha)
        if not compute_grad:
            return K
        dK = np.zeros((cov_N, N, N))
        with np.errstate(all="ignore"):
            for i in range(D):  # :349-357
                Ki = squareform(
                    pdist(np.reshape(1.0 / ell[i] * X[:, i], (-1, 1)), "sqeuclidean")
                )
                dK[i] = sf2 * M ** (-alpha - 1) * Ki
        dK[D] = 2 * K
        dK[D + 1] = K * (0.5 * tmp / M - alpha * np.log(M))  # :363
        return K, dK.transpose(1, 2, 0)

    raise ValueError(kernel)


def covariance_planes(kernel: str, hyp: np.ndarray, X: np.ndarray, degree: int = 0):
    """Generator: first K, then dK[:, :, 0], dK[:, :, 1], ... -- each plane a fresh C-contiguous (N, N) array
    holding exactly the values ``covariance(..., compute_grad=True)`` puts into that plane (the same SciPy calls
    on the same arguments; factors that the reference recomputes inside its loop over dimensions --
    ``sf2 * (df(tmp) * exp(-tmp))`` at covariance_functions.py:280, ``sf2 * M ** (-alpha - 1)`` at :357 -- are
    evaluated once: same operands, same operations, same bits).  Never more than one plane alive, so the
    gradient of a problem whose (N, N, cov_N) tensor does not fit in memory can still be formed with the
    reference's arithmetic.  Pinned plane by plane against cov_cases.npz (tests/test_oracle_golden.py)."""
    N, D = X.shape
    cov_N = cov_count(kernel, D)
    if hyp.size != cov_N or hyp.ndim != 1:
        raise ValueError("covariance_planes: one hyperparameter vector of the kernel's size")
    iso = kernel.endswith("_iso")
    if iso:
        ell = np.exp(hyp[0])
        sf2 = np.exp(2 * hyp[1])
    else:
        ell = np.exp(hyp[0:D])
        sf2 = np.exp(2 * hyp[D])

    def one_dim(scale_i, i):  # the per-dimension squared distance of :179-181, :268-275, :350-355
        return squareform(pdist(np.reshape(scale_i * X[:, i], (-1, 1)), "sqeuclidean"))

    if kernel in ("se", "se_iso"):
        tmp = squareform(pdist(X / ell, "sqeuclidean"))
        K = sf2 * np.exp(-tmp / 2)
        yield K
        if iso:
            yield K * tmp  # isotropic :216 recomputes the same squareform(pdist(X / ell))
        else:
            del tmp
            for i in range(D):  # :177-181 divides: X[:, i] / ell[i]
                yield K * squareform(pdist(np.reshape(X[:, i] / ell[i], (-1, 1)), "sqeuclidean"))
        yield 2 * K
        return
    if kernel in ("matern", "matern_iso"):
        f, df = _matern_f_df(degree)
        if iso:
            tmp = squareform(pdist(X * np.sqrt(degree) / ell))
        else:
            tmp = squareform(pdist(X @ np.diag(np.sqrt(degree) / ell)))
        K = sf2 * f(tmp) * np.exp(-tmp)
        yield K
        with np.errstate(all="ignore"):
            G = sf2 * (df(tmp) * np.exp(-tmp))
            del tmp
            if iso:
                yield G * squareform(pdist(np.sqrt(degree) / ell * X, "sqeuclidean"))
            else:
                for i in range(D):
                    yield G * one_dim(np.sqrt(degree) / ell[i], i)
            del G
        yield 2 * K
        return
    if kernel == "rq":
        alpha = np.exp(hyp[D + 1])
        tmp = squareform(pdist(X @ np.diag(1.0 / ell), "sqeuclidean"))
        M = 1 + 0.5 * tmp / alpha
        K = sf2 * M ** (-alpha)
        yield K
        with np.errstate(all="ignore"):
            G = sf2 * M ** (-alpha - 1)
            for i in range(D):
                yield G * one_dim(1.0 / ell[i], i)
            del G
        yield 2 * K
        yield K * (0.5 * tmp / M - alpha * np.log(M))
        return
    raise ValueError(kernel)


# --------------------------------------------------------------------------
# noise and mean (boundary plugins, O(N*D))
# --------------------------------------------------------------------------


def noise(params, hyp, X, y, s2=None, compute_grad=False):
    """noise_functions.py:227-283.  Returns a Python/NumPy scalar when there is
    no per-point term, else an (N,1) array -- the distinction drives the branch
    at gaussian_process.py:2407 / :2491."""
    N = X.shape[0]
    noise_N = noise_count(params)
    if hyp.size != noise_N:
        raise ValueError(
            f"Expected {noise_N} noise function hyperparameters, "
            f"{hyp.size} passed instead."
        )
    dsn2 = None
    if compute_grad:
        if any(p > 0 for p in params[1:]):
            dsn2 = np.zeros((N, noise_N))
        else:
            dsn2 = np.zeros((1, noise_N))
    i = 0
    if params[0] == 0:
        sn2 = np.spacing(1.0)
    else:
        sn2 = np.exp(2 * hyp[i])
        if compute_grad:
            dsn2[:, i] = 2 * sn2
        i += 1
    if s2 is None:
        s2 = 0
    if params[1] == 1:
        sn2 = sn2 + s2
    elif params[1] == 2:
        sn2 = sn2 + np.exp(hyp[i]) * s2
        if compute_grad:
            dsn2[:, i : i + 1] = np.exp(hyp[i]) * s2
        i += 1
    if params[2] == 1:
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


def mean(kind: str, hyp, X, compute_grad=False):
    """mean_functions.py:82-131 (zero), :210-260 (const), :340-397 (negquad)."""
    N, D = X.shape
    mean_N = mean_count(kind, D)
    if hyp.size != mean_N:
        raise ValueError(
            f"Expected {mean_N} mean function hyperparameters, "
            f"{hyp.size} passed instead."
        )
    if kind == "zero":
        m = np.zeros((N,))
        return (m, []) if compute_grad else m
    if kind == "const":
        m = hyp[0] * np.ones((N,))
        return (m, np.ones((N, 1))) if compute_grad else m
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


# --------------------------------------------------------------------------
# core computation (gaussian_process.py:2357-2521)
# --------------------------------------------------------------------------


class OraclePosterior:
    """Field-for-field the reference's Posterior record (:2568-2586)."""

    def __init__(self, hyp, alpha, sW, L, sn2_mult, L_chol):
        self.hyp = hyp
        self.alpha = alpha
        self.sW = sW
        self.L = L
        self.sn2_mult = sn2_mult
        self.L_chol = L_chol


def core(model, hyp, X, y, s2, compute_nlZ, compute_nlZ_grad, force_mult=None, streamed=False):
    """model = dict(kernel=..., degree=..., mean=..., noise=(c,u,r)).

    Returns nlZ | (nlZ, dnlZ) | OraclePosterior exactly like the reference.

    ``streamed`` (see ``core_streamed``): the covariance gradient is contracted plane by plane from
    ``covariance_planes`` instead of from the materialised (N, N, cov_N) tensor; everything else is this very
    code path.

    ``force_mult`` (test-only, not in the reference): start -- and stay -- at this jitter
    multiplier instead of escalating from 1 (:2413-2421), so that the arithmetic after a
    retry can be compared at the level the DEVICE settled on when its first successful
    level differs from LAPACK's (the level at which a near-singular Cholesky first
    succeeds is rounding dependent).  Raises LinAlgError if LAPACK fails at that level.
    """
    N, d = X.shape
    kernel, degree = model["kernel"], model.get("degree", 0)
    cov_N = cov_count(kernel, d)
    mean_N = mean_count(model["mean"], d)
    noise_N = noise_count(model["noise"])
    h_cov = hyp[0:cov_N]
    h_noise = hyp[cov_N : cov_N + noise_N]
    h_mean = hyp[cov_N + noise_N : cov_N + noise_N + mean_N]

    if compute_nlZ_grad:
        sn2, dsn2 = noise(model["noise"], h_noise, X, y, s2, compute_grad=True)
        m, dm = mean(model["mean"], h_mean, X, compute_grad=True)
        m = m.reshape((-1, 1))
        if streamed:
            planes = covariance_planes(kernel, h_cov, X, degree=degree)
            K = next(planes)
        else:
            K, dK = covariance(kernel, h_cov, X, compute_grad=True, degree=degree)
    else:
        sn2 = noise(model["noise"], h_noise, X, y, s2)
        m = np.reshape(mean(model["mean"], h_mean, X), (-1, 1))
        K = covariance(kernel, h_cov, X, degree=degree)
    sn2_mult = 1 if force_mult is None else force_mult
    tries = 10 if force_mult is None else 1

    L_chol = np.min(sn2) >= 1e-6  # :2404
    L = None
    if L_chol:
        if np.isscalar(sn2):
            sn2_div = sn2
            sn2_mat = np.eye(N)
        else:
            sn2_div = np.min(sn2)
            sn2_mat = np.diag(sn2.ravel() / sn2_div)
        for _ in range(tries):
            try:
                L = sla.cholesky(
                    K / (sn2_div * sn2_mult) + sn2_mat, check_finite=False
                )
            except sla.LinAlgError:
                sn2_mult *= 10
                continue
            break
        sl = sn2_div * sn2_mult
        pL = L
    else:
        if np.isscalar(sn2):
            sn2_mat = sn2 * np.eye(N)
        else:
            sn2_mat = np.diag(sn2.ravel())
        for _ in range(tries):
            try:
                L = sla.cholesky(K + sn2_mult * sn2_mat, check_finite=False)
            except sla.LinAlgError:
                sn2_mult *= 10
                continue
            break
        sl = 1
        if not compute_nlZ and L is not None:
            pL = sla.solve_triangular(
                -L,
                sla.solve_triangular(L, np.eye(N), trans=1.0, check_finite=False),
                trans=0,
                check_finite=False,
            )
    if L is None:
        raise sla.LinAlgError("Singular matrix for L Cholesky decomposition")

    alpha = (
        sla.solve_triangular(
            L,
            sla.solve_triangular(L, y - m, trans=1, check_finite=False),
            trans=0,
            check_finite=False,
        )
        / sl
    )

    if compute_nlZ:
        nlZ = (
            np.dot((y - m).T, alpha / 2)
            + np.sum(np.log(np.diag(L)))
            + N * np.log(2 * np.pi * sl) / 2
        )
        if compute_nlZ_grad:
            dnlZ = np.zeros(hyp.shape)
            Q = sla.solve_triangular(
                L,
                sla.solve_triangular(L, np.eye(N), trans=1, check_finite=False),
                trans=0,
                check_finite=False,
            ) / sl - np.dot(alpha, alpha.T)
            if streamed:
                L = pL = None  # the factor is not needed past Q; K stays alive inside the generator
                for i, plane in enumerate(planes):
                    dnlZ[i] = np.sum(np.sum(Q * plane)) / 2  # :2487-2488 on a contiguous (N, N) plane,
                    del plane  # which is what dK[:, :, i] of the reference's (cov_N, N, N) C array is
                assert i == cov_N - 1
            else:
                for i in range(cov_N):
                    dnlZ[i] = np.sum(np.sum(Q * dK[:, :, i])) / 2
            if np.isscalar(sn2):
                tr_Q = np.trace(Q)
                for i in range(noise_N):
                    dnlZ[cov_N + i] = (0.5 * sn2_mult * np.dot(dsn2[i], tr_Q)).item()
            else:
                dg_Q = np.diag(Q)
                for i in range(noise_N):
                    dnlZ[cov_N + i] = 0.5 * sn2_mult * np.sum(dsn2[:, i] * dg_Q)
            if mean_N > 0:
                dnlZ[cov_N + noise_N :] = np.dot(-dm.T, alpha)[:, 0]
            return nlZ[0, 0], dnlZ
        return nlZ[0, 0]

    return OraclePosterior(
        hyp,
        alpha,
        np.ones((N, 1)) / np.sqrt(np.min(sn2) * sn2_mult),
        pL,
        sn2_mult,
        L_chol,
    )


def core_streamed(model, hyp, X, y, s2, force_mult=None):
    """(nlZ, dnlZ) of ``core`` without ever holding the (N, N, cov_N) gradient tensor: K, the Cholesky factor,
    alpha, nlZ and Q through the same SciPy calls (gaussian_process.py:2415-2417, :2455-2484), then ONE gradient
    plane at a time (covariance_functions.py:349-363 and its siblings, ``covariance_planes``), each contracted as
    :2487-2488 does.  Bit-identical to ``core`` wherever ``core`` can run (pinned on every core_cases.npz
    gradient and on cfg3 sample 0 of fullsize_cases.npz, tests/test_oracle_golden.py); at cfg4 it is the
    independent fp64 value for the gradient that the reference itself cannot produce in 64 GB."""
    return core(model, hyp, X, y, s2, 1, 1, force_mult=force_mult, streamed=True)


def posteriors(model, hyps, X, y, s2, force_mult=None):
    """GP.update full-recompute loop, gaussian_process.py:870-884."""
    hyps = np.atleast_2d(hyps)
    fm = [None] * hyps.shape[0] if force_mult is None else list(force_mult)
    return [core(model, hyps[i], X, y, s2, 0, 0, force_mult=fm[i]) for i in range(hyps.shape[0])]


# --------------------------------------------------------------------------
# extended-precision restatement of the same formulas (test-only yardstick)
# --------------------------------------------------------------------------


def _chol_ld(A):
    """Lower Cholesky factor in np.longdouble (x87 80-bit: eps = 1.1e-19)."""
    n = A.shape[0]
    L = np.zeros((n, n), dtype=np.longdouble)
    for j in range(n):
        d = A[j, j] - L[j, :j] @ L[j, :j]
        if not d > 0:
            raise sla.LinAlgError("not positive definite in extended precision")
        L[j, j] = np.sqrt(d)
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def _tri_inv_ld(L):
    n = L.shape[0]
    W = np.zeros((n, n), dtype=np.longdouble)
    I = np.eye(n, dtype=np.longdouble)
    for i in range(n):
        W[i, :] = (I[i, :] - L[i, :i] @ W[:i, :]) / L[i, i]
    return W


def core_extended(model, hyp, X, y, s2, sn2_mult=1, with_scale=False):
    """nlZ, dnlZ of gaussian_process.py:2404-2508 at a GIVEN jitter multiplier with the
    factorization, solves and contractions carried in np.longdouble (K, dK, mean and noise
    values are the float64 ones: they are inputs of the factorization on every
    implementation).  Also returns cond_2(A) of the factored matrix.  Small N only (pure
    NumPy loops).  Used to put the device's and LAPACK's errors on ill-conditioned fixtures
    on one scale: both are expected within a modest multiple of cond * eps_64 of this.

    With ``with_scale`` a fifth value is returned: per gradient component the magnitude of
    the summands the component is a signed sum of (sum_ij |A^-1_ij dK_ij| / 2sl + sum_ij
    |alpha_i alpha_j dK_ij| / 2 and their noise / mean analogues).  On singular fixtures the terms are ~1e10 and cancel to
    ~10: the error of any finite-precision evaluation scales with the terms, not the result."""
    ld = np.longdouble
    N, d = X.shape
    kernel, degree = model["kernel"], model.get("degree", 0)
    cov_N = cov_count(kernel, d)
    mean_N = mean_count(model["mean"], d)
    noise_N = noise_count(model["noise"])
    h_noise = hyp[cov_N: cov_N + noise_N]
    h_mean = hyp[cov_N + noise_N: cov_N + noise_N + mean_N]
    sn2, dsn2 = noise(model["noise"], h_noise, X, y, s2, compute_grad=True)
    m, dm = mean(model["mean"], h_mean, X, compute_grad=True)
    with np.errstate(all="ignore"):
        K, dK = covariance(kernel, hyp[0:cov_N], X, compute_grad=True, degree=degree)
    r = (y - m.reshape((-1, 1))).astype(ld)
    L_chol = np.min(sn2) >= 1e-6
    if L_chol:
        sn2_div = sn2 if np.isscalar(sn2) else np.min(sn2)
        dg = np.ones(N) if np.isscalar(sn2) else sn2.ravel() / sn2_div
        sl = ld(sn2_div) * ld(sn2_mult)
        A = K.astype(ld) / sl + np.diag(dg.astype(ld))
    else:
        dg = (sn2 * np.ones(N)) if np.isscalar(sn2) else sn2.ravel()
        sl = ld(1)
        A = K.astype(ld) + ld(sn2_mult) * np.diag(dg.astype(ld))
    L = _chol_ld(A)
    W = _tri_inv_ld(L)
    Ainv = W.T @ W
    alpha = (Ainv @ r) / sl
    nlZ = (r.T @ alpha / 2)[0, 0] + np.sum(np.log(np.diag(L))) + N * np.log(2 * ld(np.pi) * sl) / 2
    Q = Ainv / sl - alpha @ alpha.T
    dnlZ = np.zeros(hyp.shape, dtype=ld)
    gscale = np.zeros(hyp.shape)
    aa = alpha @ alpha.T
    with np.errstate(all="ignore"):
        for i in range(cov_N):
            dKi = dK[:, :, i].astype(ld)
            dnlZ[i] = np.sum(Q * dKi) / 2
            gscale[i] = float(np.nansum(np.abs(Ainv * dKi)) / sl + np.nansum(np.abs(aa * dKi))) / 2
    if np.isscalar(sn2):
        for i in range(noise_N):
            dnlZ[cov_N + i] = 0.5 * sn2_mult * ld(dsn2[0, i]) * np.trace(Q)
            gscale[cov_N + i] = float(0.5 * sn2_mult * abs(dsn2[0, i]) * (np.trace(Ainv) / sl + np.trace(aa)))
    else:
        for i in range(noise_N):
            dnlZ[cov_N + i] = 0.5 * sn2_mult * np.sum(dsn2[:, i].astype(ld) * np.diag(Q))
            gscale[cov_N + i] = float(0.5 * sn2_mult * np.sum(np.abs(dsn2[:, i]) * (np.diag(Ainv) / sl + np.diag(aa))))
    if mean_N > 0:
        dnlZ[cov_N + noise_N:] = (-(np.asarray(dm).astype(ld)).T @ alpha)[:, 0]
        gscale[cov_N + noise_N:] = (np.abs(np.asarray(dm)).T @ np.abs(alpha).astype(float))[:, 0]
    # 2-norm condition number of A from its float64 eigenvalues, clamped below by the
    # extended-precision estimate |A|_F |A^-1|_F / N when float64 loses the small end
    ev = np.linalg.eigvalsh(A.astype(float))
    cond = float(ev[-1] / ev[0]) if ev[0] > 0 else np.inf
    cond_f = float(np.sqrt(np.sum(A * A)) * np.sqrt(np.sum(Ainv * Ainv))) / N
    cond = max(cond, cond_f) if np.isfinite(cond) else max(cond_f, 1.0)
    if with_scale:
        return float(nlZ), dnlZ.astype(float), cond, bool(L_chol), gscale
    return float(nlZ), dnlZ.astype(float), cond, bool(L_chol)


# --------------------------------------------------------------------------
# rank-one update (gaussian_process.py:750-844, :866-869)
# --------------------------------------------------------------------------


def rank_one_update(model, posts, X, y, x_new, y_new):
    """GP.update with ONE new point and no new hyperparameters: every posterior gets a new last
    row/column (high-noise: of the Cholesky factor; low-noise: of -inv) and an updated alpha; a
    posterior whose update is numerically unstable (sqrt_arg <= 0, :789-798) is recomputed from
    scratch on the extended data (:866-869).  Returns (posts, X_ext, y_ext, full_updates)."""
    kernel, degree = model["kernel"], model.get("degree", 0)
    d = X.shape[1]
    cov_N = cov_count(kernel, d)
    noise_N = noise_count(model["noise"])
    m_star, v_star = predict(model, posts, X, y, x_new, y_new, add_noise=True, separate_samples=True)
    full_updates = []
    for s, p in enumerate(posts):
        hyp_s = p.hyp
        sn2 = noise(model["noise"], hyp_s[cov_N: cov_N + noise_N], x_new, y_new, 0)
        sn2_eff = sn2 * p.sn2_mult
        K = covariance(kernel, hyp_s[0:cov_N], x_new, degree=degree)
        Ks = covariance(kernel, hyp_s[0:cov_N], X, x_new, degree=degree)
        L = p.L
        if p.L_chol:
            new_col = sla.solve_triangular(L, Ks, trans=1, check_finite=False)
            sqrt_arg = sn2_eff**2 + K * sn2_eff - np.dot(new_col.T, new_col)
            if sqrt_arg <= 0.0:
                full_updates.append(s)
                continue
            alpha_update = sla.solve_triangular(L, new_col, trans=0, check_finite=False) / sn2_eff
            p.L = np.block([[L, new_col / sn2_eff], [np.zeros((1, L.shape[0])), np.sqrt(sqrt_arg) / sn2_eff]])
        else:
            alpha_update = np.dot(-L, Ks)
            v = -alpha_update / v_star[:, s]
            p.L = np.block([[L + np.dot(v, alpha_update.T), -v], [-v.T, -1 / v_star[:, s]]])
        p.sW = np.concatenate((p.sW, np.array([[1 / np.sqrt(sn2_eff)]])))
        p.alpha = np.concatenate((p.alpha, np.array([[0]]))) + (m_star[:, s] - y_new) / v_star[:, s] * np.concatenate(
            (alpha_update, np.array([[-1]]))
        )
    X2, y2 = np.concatenate((X, x_new)), np.concatenate((y, y_new))
    for s in full_updates:
        posts[s] = core(model, posts[s].hyp, X2, y2, None, 0, 0)
    return posts, X2, y2, full_updates


# --------------------------------------------------------------------------
# predict (gaussian_process.py:1663-1816)
# --------------------------------------------------------------------------


def predict(
    model,
    posts,
    X,
    y,
    x_star,
    y_star=None,
    s2_star=None,
    add_noise=False,
    separate_samples=False,
    return_lpd=False,
):
    s_N = len(posts)
    N_star, D = x_star.shape
    kernel, degree = model["kernel"], model.get("degree", 0)
    cov_N = cov_count(kernel, D)
    mean_N = mean_count(model["mean"], D)
    noise_N = noise_count(model["noise"])
    mu = np.zeros((N_star, s_N))
    s2 = np.zeros((N_star, s_N))
    if return_lpd:
        if y_star is None:
            raise ValueError(
                "Cannot calculate log predictive density without y_star."
            )
        if separate_samples:
            lpd = np.zeros((N_star, s_N))
    if return_lpd or add_noise:
        y_s2 = np.zeros((N_star, s_N))

    for s in range(s_N):
        p = posts[s]
        hyp, alpha, L, L_chol, sW = p.hyp, p.alpha, p.L, p.L_chol, p.sW
        m_star = np.reshape(
            mean(model["mean"], hyp[cov_N + noise_N : cov_N + noise_N + mean_N], x_star),
            (-1, 1),
        )
        kss = covariance(kernel, hyp[0:cov_N], x_star, compute_diag=True, degree=degree)
        if y is not None:
            Ks = covariance(kernel, hyp[0:cov_N], X, x_star, degree=degree)
            mu[:, s : s + 1] = m_star + np.dot(Ks.T, alpha)
            if L_chol:
                V = sla.solve_triangular(
                    L, np.tile(sW, (1, N_star)) * Ks, trans=1, check_finite=False
                )
                s2[:, s : s + 1] = kss - np.reshape(np.sum(V * V, 0), (-1, 1))
            else:
                s2[:, s : s + 1] = kss + np.reshape(
                    np.sum(Ks * np.dot(L, Ks), 0), (-1, 1)
                )
        else:
            mu[:, s : s + 1] = m_star
            s2[:, s : s + 1] = kss
        s2[:, s] = np.maximum(s2[:, s], 0)
        if return_lpd or add_noise:
            sn2_mult = p.sn2_mult if p.sn2_mult is not None else 1
            sn2_star = noise(
                model["noise"], hyp[cov_N : cov_N + noise_N], x_star, y_star, s2_star
            )
            y_s2[:, s : s + 1] = s2[:, s : s + 1] + sn2_star * sn2_mult
        if return_lpd and separate_samples:
            lpd[:, s : s + 1] = -0.5 * (y_star - mu[:, s : s + 1]) ** 2 / y_s2[
                :, s : s + 1
            ] - 0.5 * np.log(2 * np.pi * y_s2[:, s : s + 1])

    if add_noise:
        s2 = y_s2
    if not separate_samples:
        if s_N > 1:
            mu_bar = np.reshape(np.sum(mu, 1), (-1, 1)) / s_N
            v = np.sum((mu - mu_bar) ** 2, 1) / (s_N - 1)
            s2 = np.reshape(np.sum(s2, 1) / s_N + v, (-1, 1))
            mu = mu_bar
        else:
            v = 0
        if return_lpd and add_noise:
            lpd = -0.5 * (y_star - mu) ** 2 / s2 - 0.5 * np.log(2 * np.pi * s2)
        elif return_lpd:
            y_s2 = np.reshape(np.sum(y_s2, 1) / s_N + v, (-1, 1))
            lpd = -0.5 * (y_star - mu) ** 2 / y_s2 - 0.5 * np.log(2 * np.pi * y_s2)
    if return_lpd:
        return mu, s2, lpd
    return mu, s2


# --------------------------------------------------------------------------
# the synthetic workload of SURVEY.md section 8(d) / BASELINE.md section 3
# --------------------------------------------------------------------------

BENCH_CONFIGS = {
    2: dict(N=2048, D=5, kernel="se", degree=0, S=1),
    3: dict(N=4096, D=10, kernel="matern", degree=5, S=16),
    4: dict(N=16384, D=20, kernel="rq", degree=0, S=1),
    5: dict(N=8192, D=8, kernel="se", degree=0, S=64),
    # (not a BASELINE configuration: twice the N that the device library accepted before round 6 -- the reference
    # factorizes whatever fits host memory, gaussian_process.py:2415-2417, :2477-2484)
    6: dict(N=32768, D=5, kernel="se", degree=0, S=2),
}


def synthetic_problem(cfg_idx: int, N=None, D=None, kernel=None, degree=None, S=None):
    """Seeded synthetic inputs: draw order X, noise for y, then hyp."""
    c = dict(BENCH_CONFIGS.get(cfg_idx, BENCH_CONFIGS[3]))
    for k, v in dict(N=N, D=D, kernel=kernel, degree=degree, S=S).items():
        if v is not None:
            c[k] = v
    N, D, S = c["N"], c["D"], c["S"]
    rng = np.random.default_rng(1000 + cfg_idx)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal(
        (N, 1)
    )
    base = [np.log(1.5 * np.sqrt(D) * (1 + 0.1 * d / D)) for d in range(D)]
    base.append(0.0)  # log sigma_f
    if c["kernel"] == "rq":
        base.append(0.0)  # log alpha
    if c["kernel"].endswith("_iso"):
        base = [np.log(1.5 * np.sqrt(D)), 0.0]
    base += [np.log(0.1), 0.0]  # noise, mean
    base = np.asarray(base)
    hyp = base + 0.1 * rng.standard_normal((S, base.size))
    model = dict(kernel=c["kernel"], degree=c["degree"], mean="const", noise=(1, 0, 0))
    return model, X, y, hyp

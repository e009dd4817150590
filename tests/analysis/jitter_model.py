"""Why does the device need more jitter than LAPACK on singular matrices?  CPU model (NumPy, fp64) of the leaf's
arithmetic (gpyreg_amd/csrc/leaf.h) on the four singular fixtures of tests/golden/core_cases.npz, one variant per
suspect, against scipy.linalg.cholesky:

  lapack        scipy.linalg.cholesky (the reference's call)
  scalar        right-looking scalar Cholesky, exact sqrt and division
  scalar_rsq    ... pivot via r = 1/sqrt(s) (correctly rounded here), column scaled by MULTIPLYING with r
  panel_solve   16-wide panels; panel solve L_iP = A_iP L_PP^-T by forward substitution (backward stable)
  panel_inv     16-wide panels; panel solve as the PRODUCT A_iP W_PP^T with the explicit inverse of the 16 x 16
                diagonal block (what leaf.h does)
  panel_inv_ref panel_inv + one step of refinement of the product: L += (A_iP - L L_PP^T) W_PP^T
  device_fast   the device's plan (tests/blocked_model.py, 128-tiles, trsm as a product with the explicit inverse at
                every node) with the leaf's arithmetic (= panel_inv inside the leaves)
  device_stable the same in the device's STABLE mode: one refinement step in the leaf's panel solve and at every node

Prints, per fixture sample, the first jitter multiplier 10^k at which each variant succeeds.  usage: python tests/analysis/jitter_model.py"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import parse_core_name  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402  (under tests/: the oracle is test infrastructure)


def system(model, hyp, X, y, s2, mult):
    d = X.shape[1]
    cov_N, noise_N = orc.cov_count(model["kernel"], d), orc.noise_count(model["noise"])
    sn2 = orc.noise(model["noise"], hyp[cov_N:cov_N + noise_N], X, y, s2)
    K = orc.covariance(model["kernel"], hyp[:cov_N], X, degree=model.get("degree", 0))
    N = X.shape[0]
    if np.min(sn2) >= 1e-6:
        div = sn2 if np.isscalar(sn2) else np.min(sn2)
        return K / (div * mult) + (np.eye(N) if np.isscalar(sn2) else np.diag(sn2.ravel() / div))
    return K + mult * (sn2 * np.eye(N) if np.isscalar(sn2) else np.diag(sn2.ravel()))


def chol_scalar(A, rsq=False):
    L = np.tril(A).copy()
    n = L.shape[0]
    for j in range(n):
        s = L[j, j]
        if not s > 0:
            return None
        if rsq:
            r = 1.0 / np.sqrt(s)
            L[j:, j] *= r
        else:
            L[j, j] = np.sqrt(s)
            L[j + 1:, j] /= L[j, j]
        for k in range(j + 1, n):
            L[k:, k] -= L[k:, j] * L[k, j]
    return L


def chol_panel(A, mode, pw=16):
    n = A.shape[0]
    npad = -(-n // pw) * pw
    S = np.eye(npad)
    S[:n, :n] = A
    L = np.zeros_like(S)
    for p in range(0, npad, pw):
        P = slice(p, p + pw)
        Lpp = chol_scalar(S[P, P], rsq=True)
        if Lpp is None:
            return None
        L[P, P] = Lpp
        if p + pw < npad:
            R = slice(p + pw, npad)
            if mode == "solve":
                Lr = sla.solve_triangular(Lpp, S[R, P].T, lower=True).T
            else:
                Wpp = sla.solve_triangular(Lpp, np.eye(pw), lower=True)
                Lr = S[R, P] @ Wpp.T
                if mode == "inv_ref":
                    Lr = Lr + (S[R, P] - Lr @ Lpp.T) @ Wpp.T
            L[R, P] = Lr
            S[R, R] -= Lr @ Lr.T
    return L[:n, :n]


def device_plan(A, stable):
    """The whole device algorithm on the CPU (tests/blocked_model.py: plan.h with NumPy tiles of 128 and the leaf's
    16-wide panel arithmetic), fast or stable mode."""
    import blocked_model as bm

    P = bm.pad_identity(A, 128)
    n = P.shape[0]
    W, T = np.zeros((n, n)), np.zeros((n, n))
    info = bm.potrf_inv(P, W, T, 0, n, 128, True, False, stable=stable,
                        leaf_fn=lambda a, w: bm.leaf_panels(a, w, 16, refine=stable))
    return None if info else P


VARIANTS = {
    "lapack": lambda A: _lapack(A),
    "scalar": lambda A: chol_scalar(A),
    "scalar_rsq": lambda A: chol_scalar(A, rsq=True),
    "panel_solve": lambda A: chol_panel(A, "solve"),
    "panel_inv": lambda A: chol_panel(A, "inv"),
    "panel_inv_ref": lambda A: chol_panel(A, "inv_ref"),
    "device_fast": lambda A: device_plan(A, False),
    "device_stable": lambda A: device_plan(A, True),
}


def _lapack(A):
    try:
        return sla.cholesky(A, lower=True, check_finite=False)
    except sla.LinAlgError:
        return None


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "core_cases.npz"), allow_pickle=False)
    want = ("g008", "g029", "g030", "g031")
    print("%-44s " % "fixture sample" + " ".join("%13s" % v for v in VARIANTS))
    for name in g["names"]:
        tag, model, N, D, flavour = parse_core_name(name)
        if tag not in want:
            continue
        X, y = g[tag + "_X"], g[tag + "_y"]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        for s, hyp in enumerate(g[tag + "_hyp"]):
            row = []
            for v, f in VARIANTS.items():
                lvl = None
                for k in range(10):
                    with np.errstate(all="ignore"):
                        if f(system(model, hyp, X, y, s2, 10.0 ** k)) is not None:
                            lvl = k
                            break
                row.append("1e%d" % lvl if lvl is not None else "fail")
            print("%-44s " % (str(name)[:38] + " s=%d" % s) + " ".join("%13s" % r for r in row))


if __name__ == "__main__":
    main()

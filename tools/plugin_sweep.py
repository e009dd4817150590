"""Host-side plugin protocol sweep (no device): hyperparameter_count / hyperparameter_info / get_bounds_info of every
covariance, mean and noise class, and the values + gradients of every mean and noise configuration, printed for a diff
against the reference (protocol as in tools/api_sweep.py):

    GPYREG_MODULE=gpyreg PYTHONPATH=/root/reference:/root/repo python -W ignore tools/plugin_sweep.py > ref.txt
"""
import importlib
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpr = importlib.import_module(os.environ.get("GPYREG_MODULE", "gpyreg_amd"))


def arr(v):
    a = np.asarray(v, dtype=float)
    return str(a.shape) + " " + " ".join("%.10g" % x for x in a.ravel()[:24])


def bounds(tag, b):
    for k in sorted(b):
        print(tag, k, arr(b[k]))


def main():
    rng = np.random.default_rng(21)
    cov = gpr.covariance_functions
    iso = gpr.isotropic_covariance_functions
    kernels = [("se", cov.SquaredExponential()), ("m1", cov.Matern(1)), ("m3", cov.Matern(3)), ("m5", cov.Matern(5)),
               ("rq", cov.RationalQuadraticARD()), ("se_iso", iso.SquaredExponentialIsotropic()),
               ("m_iso1", iso.MaternIsotropic(1)), ("m_iso3", iso.MaternIsotropic(3)), ("m_iso5", iso.MaternIsotropic(5))]
    for N, D in ((1, 1), (7, 1), (30, 3), (12, 5)):
        X = rng.uniform(-4, 6, (N, D)) * (1 + np.arange(D))
        if N > 3:
            X[1] = X[0]  # a duplicated point (zero distances in the bounds heuristics)
        y = np.sin(X.sum(1, keepdims=True)) * 3 + rng.standard_normal((N, 1))
        s2 = 0.01 + 0.1 * rng.uniform(size=(N, 1))
        for name, k in kernels:
            tag = "%s.N%d.D%d" % (name, N, D)
            print(tag, "count", k.hyperparameter_count(D), "info", k.hyperparameter_info(D))
            try:
                bounds(tag, k.get_bounds_info(X, y))
            except Exception as e:  # noqa: BLE001
                print(tag, "bounds RAISES", type(e).__name__, str(e)[:80])
        for name, m in (("zero", gpr.mean_functions.ZeroMean()), ("const", gpr.mean_functions.ConstantMean()),
                        ("negquad", gpr.mean_functions.NegativeQuadratic())):
            tag = "%s.N%d.D%d" % (name, N, D)
            n = m.hyperparameter_count(D)
            print(tag, "count", n, "info", m.hyperparameter_info(D))
            try:
                bounds(tag, m.get_bounds_info(X, y))
            except Exception as e:  # noqa: BLE001
                print(tag, "bounds RAISES", type(e).__name__, str(e)[:80])
            h = 0.3 * rng.standard_normal(n)
            v, dv = m.compute(h, X, compute_grad=True)
            print(tag, "value", arr(v), "grad", arr(dv) if np.size(dv) else "[]")
            print(tag, "value only", arr(m.compute(h, X)))
        for c, u, sc, r in itertools.product((False, True), repeat=4):
            nf = gpr.noise_functions.GaussianNoise(constant_add=c, user_provided_add=u, scale_user_provided=sc,
                                                   rectified_linear_output_dependent_add=r)
            tag = "noise%d%d%d%d.N%d.D%d" % (c, u, sc, r, N, D)
            n = nf.hyperparameter_count()
            print(tag, "count", n, "info", nf.hyperparameter_info(), "parameters", arr(nf.parameters))
            try:
                bounds(tag, nf.get_bounds_info(X, y))
            except Exception as e:  # noqa: BLE001
                print(tag, "bounds RAISES", type(e).__name__, str(e)[:80])
            h = 0.3 * rng.standard_normal(n) - 1.0
            for lab, s2v in (("s2", s2), ("none", None)):
                try:
                    v, dv = nf.compute(h, X, y, s2v, compute_grad=True)
                    print(tag, lab, "scalar" if np.isscalar(v) else "array", arr(v), "grad", arr(dv))
                    print(tag, lab, "value only", arr(nf.compute(h, X, y, s2v)))
                except Exception as e:  # noqa: BLE001
                    print(tag, lab, "RAISES", type(e).__name__, str(e)[:80])


if __name__ == "__main__":
    main()

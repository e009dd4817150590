"""Numerics sweep through the public API: every covariance class x mean x noise configuration (with and without
user-provided noise, the eps-noise GP, noise below 1e-6: the reference's unscaled branch) on a small problem --
nlZ and gradient (_GP__compute_nlZ), predict with / without noise, log predictive density, separate samples,
predict_full, log_likelihood, and Bayesian quadrature for the squared-exponential kernels -- printed for a diff against
the reference (protocol as in tools/api_sweep.py):

    GPYREG_MODULE=gpyreg PYTHONPATH=/root/reference:/root/repo python -W ignore tools/numerics_sweep.py > ref.txt
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpr = importlib.import_module(os.environ.get("GPYREG_MODULE", "gpyreg_amd"))


def arr(v):
    a = np.asarray(v, dtype=float)
    fin = a[np.isfinite(a)]
    return "%s nan%d sum %.9g absmax %.9g first %s" % (a.shape, int(np.isnan(a).sum()), fin.sum() if fin.size else 0.0,
                                                      np.abs(fin).max() if fin.size else 0.0,
                                                      " ".join("%.9g" % x for x in a.ravel()[:3]))


def attempt(tag, f):
    try:
        v = f()
        if isinstance(v, tuple):
            print(tag, " | ".join(arr(x) for x in v))
        else:
            print(tag, arr(v))
    except Exception as e:  # noqa: BLE001
        print(tag, "RAISES", type(e).__name__, str(e)[:80])


def main():
    cov, iso = gpr.covariance_functions, gpr.isotropic_covariance_functions
    kernels = [("se", cov.SquaredExponential), ("m1", lambda: cov.Matern(1)), ("m3", lambda: cov.Matern(3)),
               ("m5", lambda: cov.Matern(5)), ("rq", cov.RationalQuadraticARD),
               ("se_iso", iso.SquaredExponentialIsotropic), ("m_iso1", lambda: iso.MaternIsotropic(1)),
               ("m_iso3", lambda: iso.MaternIsotropic(3)), ("m_iso5", lambda: iso.MaternIsotropic(5))]
    means = [("zero", gpr.mean_functions.ZeroMean), ("const", gpr.mean_functions.ConstantMean),
             ("negquad", gpr.mean_functions.NegativeQuadratic)]
    # (constant, user_provided, scale_user_provided, rectified), log noise scale (None: the kernel's default draw)
    noises = [((1, 0, 0, 0), None), ((1, 0, 0, 0), -8.0), ((0, 0, 0, 0), None), ((1, 1, 0, 0), None),
              ((0, 1, 1, 0), None), ((1, 0, 0, 1), None), ((0, 1, 0, 1), None)]
    N, D, S, M = int(os.environ.get("SWEEP_N", "24")), 2, 2, 5  # (300: the blocked recursion, three levels, odd splits)
    rng = np.random.default_rng(2024)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    s2 = 0.01 + 0.05 * rng.uniform(size=(N, 1))
    xs = rng.uniform(-3.5, 3.5, (M, D))
    ys = np.sin(xs.sum(1, keepdims=True))
    s2s = 0.02 * np.ones((M, 1))
    qm, qs = rng.uniform(-1, 1, (3, D)), 0.3 + rng.uniform(size=(3, D))
    for kname, mk in kernels:
        for mname, mm in means:
            for npar, lognoise in noises:
                c, u, sc, r = npar
                noise = gpr.noise_functions.GaussianNoise(constant_add=bool(c), user_provided_add=bool(u),
                                                          scale_user_provided=bool(sc),
                                                          rectified_linear_output_dependent_add=bool(r))
                k = mk()
                extra = {"dtype": os.environ["SWEEP_DTYPE"]} if os.environ.get("SWEEP_DTYPE") else {}  # (this package only)
                if os.environ.get("SWEEP_QUIRKS"):  # (this package only) the reference's numbers where its code is wrong
                    extra["reference_quirks"] = True
                gp = gpr.GP(D=D, covariance=k, mean=mm(), noise=noise, **extra)
                cov_N, noise_N = k.hyperparameter_count(D), noise.hyperparameter_count()
                mean_N = gp.mean.hyperparameter_count(D)
                hyp = np.zeros((S, cov_N + noise_N + mean_N))
                for s in range(S):
                    h_cov = 0.2 * rng.standard_normal(cov_N)
                    h_cov[:min(D, cov_N - 1)] += np.log(1.5)
                    h_noise = []
                    if c:
                        h_noise.append((np.log(0.1) if lognoise is None else lognoise) + 0.1 * rng.standard_normal())
                    if u and sc:
                        h_noise.append(0.2 * rng.standard_normal())
                    if r:
                        h_noise += [0.3 * rng.standard_normal(), np.log(0.05) + 0.1 * rng.standard_normal()]
                    h_mean = [] if mname == "zero" else ([0.2 * rng.standard_normal()] if mname == "const" else
                                                         [0.2 * rng.standard_normal()] + list(0.5 * rng.standard_normal(D))
                                                         + list(np.log(4.0) + 0.2 * rng.standard_normal(D)))
                    hyp[s] = np.concatenate([h_cov, h_noise, h_mean])
                uses_s2 = bool(u or sc)
                tag = "%s.%s.n%d%d%d%d%s" % (kname, mname, c, u, sc, r, "" if lognoise is None else "lo")
                try:
                    gp.update(X_new=X, y_new=y, s2_new=s2 if uses_s2 else None, hyp=hyp)
                except Exception as e:  # noqa: BLE001
                    print(tag, "update RAISES", type(e).__name__, str(e)[:80])
                    continue
                print(tag, "flags", [(int(p.L_chol), float(p.sn2_mult)) for p in gp.posteriors])
                attempt(tag + " nlz", lambda: gp._GP__compute_nlZ(hyp[0], True, False))
                attempt(tag + " nlz1", lambda: np.array(gp._GP__compute_nlZ(hyp[1], False, False)))
                st = s2s if uses_s2 else None
                attempt(tag + " pred", lambda: gp.predict(xs, s2_star=st, add_noise=False))
                attempt(tag + " pred_noise", lambda: gp.predict(xs, s2_star=st, add_noise=True))
                attempt(tag + " lpd", lambda: gp.predict(xs, ys, st, add_noise=True, return_lpd=True))
                attempt(tag + " sep", lambda: gp.predict(xs, ys, st, add_noise=True, separate_samples=True,
                                                          return_lpd=True))
                attempt(tag + " full", lambda: gp.predict_full(xs, s2_star=st, add_noise=True))
                attempt(tag + " loglik", lambda: np.array(gp.log_likelihood(hyp[1])))
                attempt(tag + " alpha", lambda: (gp.posteriors[1].alpha, gp.posteriors[1].sW, np.asarray(gp.posteriors[1].L)))
                if kname in ("se", "se_iso"):
                    attempt(tag + " quad", lambda: gp.quad(qm, qs, compute_var=True, separate_samples=True))
                    attempt(tag + " quad_avg", lambda: gp.quad(qm, qs, compute_var=True))


if __name__ == "__main__":
    main()

"""Seeded GP.fit runs over the option edge cases the reference's tests visit (opts_N / n_samples / init_N = 0 or 1 and
their combinations, fixed and partly fixed bounds, bounds set before the fit, user-provided noise, recommended bounds
afterwards), printed as rounded numbers for a diff against the reference (see tools/api_sweep.py for the protocol):

    GPYREG_MODULE=gpyreg PYTHONPATH=/root/reference:/root/repo python -W ignore tools/fit_sweep.py > ref.txt
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpr = importlib.import_module(os.environ.get("GPYREG_MODULE", "gpyreg_amd"))


def line(tag, *vals):
    out = []
    for v in vals:
        if v is None:
            out.append("None")
        elif isinstance(v, np.ndarray):
            out.append(str(v.shape) + " " + " ".join("%.4f" % x for x in np.asarray(v, dtype=float).ravel()[:12]))
        else:
            out.append("%.4f" % float(v))
    print(tag, " | ".join(out))


def main():
    N, D = 20, 1
    X = np.reshape(np.linspace(-10, 10, N), (-1, 1))
    y = 1 + np.sin(X) + 0.1 * np.random.default_rng(5).standard_normal((N, 1))  # (noise-free data make the fit chaotic)
    mk = lambda: gpr.GP(D=D, covariance=gpr.covariance_functions.SquaredExponential(),
                        mean=gpr.mean_functions.ConstantMean(),
                        noise=gpr.noise_functions.GaussianNoise(constant_add=True))
    xs = np.linspace(-12, 12, 5).reshape(-1, 1)
    gp = mk()
    option_sets = [{"opts_N": 0}, {"n_samples": 0}, {"init_N": 0}, {"opts_N": 0, "n_samples": 0},
                   {"n_samples": 0, "init_N": 0}, {"opts_N": 0, "init_N": 0},
                   {"opts_N": 0, "n_samples": 0, "init_N": 0}, {"init_N": 1},
                   {"n_samples": 4, "thin": 1, "burn": 3, "init_N": 64}]
    for k, opts in enumerate(option_sets):  # in a row on ONE object, as the reference's test does
        np.random.seed(300 + k)
        try:
            hyp, res, samp = gp.fit(X=X, y=y, options=dict(opts))
            line("opt%d.hyp" % k, hyp)
            line("opt%d.res" % k, None if res is None else res.fun, None if res is None else np.asarray(res.x))
            line("opt%d.pred" % k, *gp.predict(xs))
            line("opt%d.bounds" % k, gp.lower_bounds, gp.upper_bounds)
        except Exception as e:  # noqa: BLE001
            print("opt%d RAISES" % k, type(e).__name__, str(e)[:80])
    # bounds fixed before the fit: all of them, then only the noise
    for k, fixed in enumerate((("covariance_log_lengthscale", "covariance_log_outputscale", "noise_log_scale", "mean_const"),
                               ("noise_log_scale",))):
        g = mk()
        b = {"covariance_log_lengthscale": (-2.0, 3.0), "covariance_log_outputscale": (-3.0, 3.0),
             "noise_log_scale": (-6.0, 1.0), "mean_const": (-2.0, 4.0)}
        vals = {"covariance_log_lengthscale": 0.7, "covariance_log_outputscale": 0.1, "noise_log_scale": -3.0,
                "mean_const": 1.0}
        for name in fixed:
            b[name] = (vals[name], vals[name])
        g.set_bounds(b)
        np.random.seed(400 + k)
        try:
            hyp, res, _ = g.fit(X=X, y=y, options={"n_samples": 3, "thin": 1, "burn": 2, "init_N": 32})
            line("fixed%d.hyp" % k, hyp)
            line("fixed%d.pred" % k, *g.predict(xs))
            line("fixed%d.bounds" % k, g.lower_bounds, g.upper_bounds)
            rb = g.get_recommended_bounds()
            line("fixed%d.recommended" % k, *[np.asarray(rb[n]).ravel() for n in sorted(rb)])
        except Exception as e:  # noqa: BLE001
            print("fixed%d RAISES" % k, type(e).__name__, str(e)[:80])
    # user-provided noise and a second fit on more data (hyperparameters carried over as a start)
    g = gpr.GP(D=D, covariance=gpr.covariance_functions.Matern(3), mean=gpr.mean_functions.NegativeQuadratic(),
               noise=gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True))
    s2 = 0.01 + 0.02 * np.abs(np.cos(X))
    np.random.seed(500)
    hyp, res, _ = g.fit(X=X[:12], y=y[:12], s2=s2[:12], options={"n_samples": 3, "thin": 1, "burn": 2, "init_N": 48})
    line("s2fit.hyp", hyp)
    np.random.seed(501)
    hyp, res, _ = g.fit(X=X, y=y, s2=s2, options={"n_samples": 3, "thin": 1, "burn": 2, "init_N": 48})
    line("s2fit.hyp2", hyp)
    line("s2fit.pred", *g.predict(xs, s2_star=0.02 * np.ones((5, 1)), add_noise=True))
    rb = g.get_recommended_bounds()
    line("s2fit.recommended", *[np.asarray(rb[n]).ravel() for n in sorted(rb)])
    # the remaining options on a 2-D problem: random design, caller's widths, bounds passed as options, df_base with
    # Student-t priors, tolerances, an unknown sampler
    rng = np.random.default_rng(9)
    X2 = rng.uniform(-3, 3, (40, 2))
    y2 = np.sin(X2.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((40, 1))
    mk2 = lambda: gpr.GP(D=2, covariance=gpr.covariance_functions.Matern(5), mean=gpr.mean_functions.ConstantMean(),
                         noise=gpr.noise_functions.GaussianNoise(constant_add=True))
    xs2 = rng.uniform(-3, 3, (4, 2))
    base = {"n_samples": 3, "thin": 1, "burn": 2, "init_N": 32}
    variants = [dict(base, init_method="rand"),
                dict(base, widths=np.array([0.3, 0.3, 0.2, 0.2, 0.5])),
                dict(base, lower_bounds=np.array([-2.0, -2.0, -3.0, -6.0, -2.0]),
                     upper_bounds=np.array([3.0, 3.0, 3.0, 1.0, 2.0])),
                dict(base, df_base=3),
                dict(base, tol_opt=1e-3, tol_opt_mcmc=1e-2),
                dict(base, step_out=True),
                dict(base, adaptive=False, n_samples=4),
                dict(base, sampler="laplace"),
                dict(base, sampler="nuts")]
    for k, opts in enumerate(variants):
        g = mk2()
        pri = g.get_priors()
        pri["noise_log_scale"] = ("student_t", (np.log(1e-2), 1.0, np.nan if k == 3 else 5.0))
        try:
            g.set_priors(pri)
            np.random.seed(600 + k)
            hyp, res, _ = g.fit(X=X2, y=y2, options=opts)
            line("more%d.hyp" % k, hyp)
            line("more%d.res" % k, res.fun, np.asarray(res.x))
            line("more%d.pred" % k, *g.predict(xs2))
            line("more%d.bounds" % k, g.lower_bounds, g.upper_bounds)
            line("more%d.df" % k, g.hyper_priors["df"])
        except Exception as e:  # noqa: BLE001
            print("more%d RAISES" % k, type(e).__name__, str(e)[:80])


if __name__ == "__main__":
    main()

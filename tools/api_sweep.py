"""API conformance sweep: the same calls, in the unusual states the reference's tests visit (a GP without data,
getters / setters, recommended bounds, error messages, shape conversion, split updates), printed as rounded numbers
so that the output of this package (GPU box) can be diffed against the reference's (build container):

    GPYREG_MODULE=gpyreg PYTHONPATH=/root/reference python tools/api_sweep.py > ref.txt      # here
    python tools/api_sweep.py > mine.txt                                                     # GPU box
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpr = importlib.import_module(os.environ.get("GPYREG_MODULE", "gpyreg_amd"))
np.set_printoptions(precision=7, suppress=True, linewidth=200)


def show(tag, v):
    if isinstance(v, tuple):
        for i, x in enumerate(v):
            show("%s[%d]" % (tag, i), x)
        return
    if isinstance(v, dict):
        for k in sorted(v):
            show("%s.%s" % (tag, k), v[k])
        return
    if isinstance(v, list):
        for i, x in enumerate(v):
            show("%s[%d]" % (tag, i), x)
        return
    if isinstance(v, np.ndarray):
        a = np.asarray(v, dtype=float)
        print(tag, a.shape, "sum %.8g absmax %.8g first %s" % (np.nansum(a), np.nanmax(np.abs(a)) if a.size else 0.0,
                                                              np.round(a.ravel()[:4], 7)))
    else:
        print(tag, repr(v) if not isinstance(v, float) else "%.10g" % v)


def attempt(tag, f):
    try:
        show(tag, f())
    except Exception as e:  # noqa: BLE001
        print(tag, "RAISES", type(e).__name__, str(e)[:90])


def main():
    D = 2
    mk = lambda: gpr.GP(D=D, covariance=gpr.covariance_functions.SquaredExponential(),
                        mean=gpr.mean_functions.NegativeQuadratic(),
                        noise=gpr.noise_functions.GaussianNoise(constant_add=True))
    gp = mk()
    show("empty.bounds", gp.get_bounds())
    show("empty.priors", gp.get_priors())
    attempt("empty.hyp", lambda: gp.get_hyperparameters(as_array=True))
    rng = np.random.default_rng(11)
    hyp = rng.standard_normal((3, 3 + 1 + 5))
    hyp[:, D] *= 0.2
    hyp[:, D + 1] *= 0.3
    gp.update(hyp=hyp)
    xs = rng.uniform(-5, 5, (7, D))
    ys = np.zeros((7, 1))
    attempt("nodata.predict_full_noise", lambda: gp.predict_full(xs, add_noise=True))
    attempt("nodata.predict_full", lambda: gp.predict_full(xs, add_noise=False))
    attempt("nodata.predict_noise", lambda: gp.predict(xs, add_noise=True))
    attempt("nodata.predict", lambda: gp.predict(xs, add_noise=False))
    attempt("nodata.lpd", lambda: gp.predict(xs, ys, return_lpd=True, add_noise=False))
    attempt("nodata.sep", lambda: gp.predict(xs, ys, return_lpd=True, add_noise=True, separate_samples=True))
    attempt("nodata.hypdict", lambda: gp.get_hyperparameters())
    attempt("nodata.recommended", lambda: gp.get_recommended_bounds())
    attempt("nodata.quad", lambda: gp.quad(0.1, 0.5))
    # data, getters and setters
    N = 25
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    gp.update(X_new=X, y_new=y, hyp=hyp)
    attempt("data.predict", lambda: gp.predict(xs, add_noise=True))
    attempt("data.lpd", lambda: gp.predict(xs, ys, return_lpd=True))
    attempt("data.recommended", lambda: gp.get_recommended_bounds())
    attempt("data.loglik", lambda: gp.log_likelihood(hyp[0]))
    attempt("data.loglik_dict", lambda: gp.log_likelihood(gp.hyperparameters_to_dict(hyp[:1])[0]))
    attempt("data.logpost", lambda: gp.log_posterior(hyp[1]))
    attempt("data.quad", lambda: gp.quad(np.zeros((2, D)), np.ones((2, D)), compute_var=True))
    hd = gp.get_hyperparameters()
    show("data.hypdict", hd)
    gp.set_hyperparameters(hd)
    attempt("data.predict_after_set", lambda: gp.predict(xs))
    gp.set_hyperparameters(hyp[0])
    attempt("data.hyp_after_1d_set", lambda: gp.get_hyperparameters(as_array=True))
    attempt("data.wrong_shape", lambda: gp.set_hyperparameters(np.zeros((2, 4))))
    attempt("data.to_dict_wrong", lambda: gp.hyperparameters_to_dict(np.zeros((2, 4))))
    attempt("data.from_dict_single", lambda: gp.hyperparameters_from_dict(hd[0]))
    attempt("data.from_dict_list", lambda: gp.hyperparameters_from_dict(hd))
    # bounds and priors round trips
    b = gp.get_bounds()
    b["noise_log_scale"] = (-7.0, 1.0)
    gp.set_bounds(b)
    show("bounds.after", gp.get_bounds())
    attempt("bounds.lower", lambda: gp.lower_bounds)
    p = gp.get_priors()
    p["mean_const"] = ("gaussian", (0.0, 2.0))
    p["noise_log_scale"] = ("student_t", (np.log(1e-3), 1.0, 7))
    p["covariance_log_lengthscale"] = ("smoothbox", (-2.0, 2.0, 0.5))
    p["covariance_log_outputscale"] = ("smoothbox_student_t", (-1.0, 1.0, 0.5, 4))
    gp.set_priors(p)
    attempt("priors.after", lambda: gp.get_priors())
    attempt("priors.logpost", lambda: gp.log_posterior(hyp[1]))
    attempt("priors.bad_name", lambda: gp.set_priors({"nonsense": None}))
    attempt("priors.bad_type", lambda: gp.set_priors(dict(p, mean_const=("cauchy", (0.0, 1.0)))))
    # split update == one update; update with only new hyperparameters; cleaning
    g1, g2 = mk(), mk()
    g1.update(X_new=X, y_new=y, hyp=hyp)
    g2.update(X_new=X[:10], y_new=y[:10], hyp=hyp)
    g2.update(X_new=X[10:], y_new=y[10:])
    show("split.diff", float(np.abs(g1.predict(xs)[0] - g2.predict(xs)[0]).max() < 1e-9))
    g2.update(hyp=hyp[:2])
    attempt("split.newhyp", lambda: g2.predict(xs, separate_samples=True))
    g2.update(X_new=xs[:3], y_new=ys[:3], compute_posterior=False)
    attempt("split.no_post_shapes", lambda: (g2.X.shape, g2.y.shape, g2.posteriors.size))
    g2.update(compute_posterior=True)
    attempt("split.after_recompute", lambda: g2.predict(xs))
    g1.clean()
    attempt("clean.alpha", lambda: g1.posteriors[0].alpha)
    attempt("clean.hyp", lambda: g1.get_hyperparameters(as_array=True))
    g1.update(compute_posterior=True)
    attempt("clean.update", lambda: g1.predict(xs))
    # shapes
    g3 = gpr.GP(D=1, covariance=gpr.covariance_functions.Matern(1), mean=gpr.mean_functions.ZeroMean(),
                noise=gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True))
    x1 = np.linspace(-2, 2, 9)
    attempt("shapes.1d", lambda: g3._convert_shapes(x1, np.sin(x1), 0.01 * np.ones(9)))
    attempt("shapes.scalar_s2", lambda: g3._convert_shapes(x1.reshape(-1, 1), np.sin(x1).reshape(-1, 1), 0.1))
    attempt("shapes.bad", lambda: g3._convert_shapes(np.zeros((3, 2)), np.zeros(3), None))
    attempt("shapes.update_1d", lambda: (g3.update(X_new=x1, y_new=np.sin(x1), s2_new=0.01 * np.ones(9),
                                                   hyp=np.array([[0.1, 0.0, np.log(0.1)]])), g3.predict(np.array([0.3]), s2_star=0.02))[1])
    # the same data in other clothes: Fortran order, float32 (exactly representable values), integers, a sliced view
    Xq = np.round(rng.uniform(-3, 3, (18, 2)) * 8) / 8
    yq = np.round(np.sin(Xq.sum(1, keepdims=True)) * 16) / 16
    hq = np.array([[0.2, 0.3, 0.1, np.log(0.2), 0.0, 0.1, -0.1, 1.0, 1.2]])
    big = np.zeros((36, 4))
    big[::2, 1:3] = Xq
    for label, Xv, yv in (("plain", Xq, yq), ("fortran", np.asfortranarray(Xq), np.asfortranarray(yq)),
                          ("float32", Xq.astype(np.float32), yq.astype(np.float32)), ("view", big[::2, 1:3], yq),
                          ("int_y", Xq, np.round(yq * 0 + 1).astype(int)), ("flat_y", Xq, yq.ravel())):
        def run(Xv=Xv, yv=yv):
            g = mk()
            g.update(X_new=Xv, y_new=yv, hyp=hq)
            return g.predict(xs, add_noise=True) + (np.array(g.log_likelihood(hq[0])),)
        attempt("clothes." + label, run)
    attempt("quad.not_se", lambda: g3.quad(0.0, 1.0))
    attempt("str", lambda: str(g3).replace("gpyreg_amd", "gpyreg"))


if __name__ == "__main__":
    main()

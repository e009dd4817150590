"""GP.update call sequences (one point, several points, with and without new hyperparameters, per-point noise,
output-dependent noise, compute_posterior off and on, points added to a GP that had none, the low-noise
parametrisation) with a prediction after every call, printed for a diff against the reference (protocol as in
tools/api_sweep.py):

    GPYREG_MODULE=gpyreg PYTHONPATH=/root/reference:/root/repo python -W ignore tools/update_sweep.py > ref.txt
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gpr = importlib.import_module(os.environ.get("GPYREG_MODULE", "gpyreg_amd"))


def arr(v):
    a = np.asarray(v, dtype=float)
    return "%s sum %.9g absmax %.9g first %s" % (a.shape, a.sum(), np.abs(a).max() if a.size else 0.0,
                                                 " ".join("%.9g" % x for x in a.ravel()[:3]))


def main():
    D = 2
    rng = np.random.default_rng(77)
    X = rng.uniform(-3, 3, (60, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((60, 1))
    s2 = 0.01 + 0.05 * rng.uniform(size=(60, 1))
    xs = rng.uniform(-3, 3, (6, D))
    configs = [("const", dict(constant_add=True), np.log(0.1)),
               ("lownoise", dict(constant_add=True), -8.0),
               ("user", dict(constant_add=True, user_provided_add=True), np.log(0.1)),
               ("rect", dict(constant_add=True, rectified_linear_output_dependent_add=True), np.log(0.1))]
    for name, kw, ln in configs:
        noise = gpr.noise_functions.GaussianNoise(**kw)
        extra = {"reference_quirks": True} if os.environ.get("SWEEP_QUIRKS") else {}  # (this package only)
        gp = gpr.GP(D=D, covariance=gpr.covariance_functions.Matern(5), mean=gpr.mean_functions.ConstantMean(),
                    noise=noise, **extra)
        nN = noise.hyperparameter_count()
        S = 3
        hyp = np.zeros((S, 3 + nN + 1))
        for s in range(S):
            h_noise = [ln + 0.1 * rng.standard_normal()] + ([0.3 * rng.standard_normal(), np.log(0.05)] if nN == 3 else [])
            hyp[s] = np.concatenate([np.log(1.5) + 0.2 * rng.standard_normal(2), [0.1 * rng.standard_normal()], h_noise,
                                     [0.2 * rng.standard_normal()]])
        use_s2 = name == "user"
        st = 0.02 * np.ones((6, 1)) if use_s2 else None

        def show(tag):
            try:
                mu, v = gp.predict(xs, s2_star=st, add_noise=True, separate_samples=True)
                fl = [(int(p.L_chol), float(p.sn2_mult)) for p in gp.posteriors]
                print(name, tag, "N", gp.X.shape[0], arr(mu), "|", arr(v), "|", fl)
            except Exception as e:  # noqa: BLE001
                print(name, tag, "RAISES", type(e).__name__, str(e)[:80])

        def step(tag, **kwargs):
            try:
                gp.update(**kwargs)
            except Exception as e:  # noqa: BLE001
                print(name, tag, "update RAISES", type(e).__name__, str(e)[:80])
                return
            show(tag)

        sl = lambda a, b: dict(X_new=X[a:b], y_new=y[a:b], **({"s2_new": s2[a:b]} if use_s2 else {}))
        step("first", hyp=hyp, **sl(0, 20))
        step("one", **sl(20, 21))
        step("one_more", **sl(21, 22))
        step("five", **sl(22, 27))
        step("one_newhyp", hyp=hyp[:2] + 0.05, **sl(27, 28))
        step("hyp_only", hyp=hyp)
        step("no_post", compute_posterior=False, **sl(28, 30))
        step("recompute", compute_posterior=True)
        step("one_after", **sl(30, 31))
        try:
            for k in range(31, 36):
                gp.update(**sl(k, k + 1))
            show("five_ones")
        except Exception as e:  # noqa: BLE001
            print(name, "five_ones update RAISES", type(e).__name__, str(e)[:80])
        step("x_only", X_new=X[36:37])
        step("bad_dim", X_new=np.zeros((1, 3)), y_new=np.zeros((1, 1)))
        # a GP that starts without data
        g0 = gpr.GP(D=D, covariance=gpr.covariance_functions.Matern(5), mean=gpr.mean_functions.ConstantMean(),
                    noise=gpr.noise_functions.GaussianNoise(**kw), **extra)
        g0.update(hyp=hyp)
        gp = g0
        show("nodata")
        step("nodata_first", **sl(0, 1))
        step("nodata_second", **sl(1, 2))
        step("nodata_more", **sl(2, 12))


if __name__ == "__main__":
    main()

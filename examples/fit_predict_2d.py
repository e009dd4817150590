"""End-to-end use of gpyreg_amd on a toy 2-D problem (the shape of the reference's
examples/example_2.py: N=20, D=2, squared-exponential kernel, constant mean, Gaussian
noise, hyperparameter priors, fit with 10 slice samples, predict on a grid, then add
points).  Needs an MI355X: every GP evaluation runs in libgpcore.so.

    python examples/fit_predict_2d.py
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpyreg_amd as gpr  # noqa: E402


def main(seed=1235, verbose=True):
    np.random.seed(seed)
    N, D = 20, 2
    X = np.random.uniform(low=-3, high=3, size=(N, D))
    y = np.reshape(np.sin(np.sum(X, 1)) + np.random.normal(scale=0.1, size=N), (-1, 1))

    gp = gpr.GP(
        D=D,
        covariance=gpr.covariance_functions.SquaredExponential(),
        mean=gpr.mean_functions.ConstantMean(),
        noise=gpr.noise_functions.GaussianNoise(constant_add=True),
    )
    gp.set_priors({
        "covariance_log_outputscale": ("student_t", (0, np.log(10), 3)),
        "covariance_log_lengthscale": ("gaussian", (np.log(np.std(X, ddof=1)), np.log(10))),
        "noise_log_scale": ("gaussian", (np.log(1e-3), 1.0)),
        "mean_const": ("smoothbox", (np.min(y), np.max(y), 1.0)),
    })
    hyp, opt_result, _ = gp.fit(X=X, y=y, options={"n_samples": 10})

    xx, yy = np.meshgrid(np.linspace(-5, 5, 20), np.linspace(-5, 5, 20))
    x_star = np.array((xx.ravel(), yy.ravel())).T
    fmu, fs2 = gp.predict(x_star, add_noise=False)

    X_new = np.random.uniform(low=-5, high=5, size=(N, D))
    y_new = np.reshape(np.sin(np.sum(X_new, 1)) + np.random.normal(scale=0.1, size=N), (-1, 1))
    gp.update(X_new=X_new, y_new=y_new)
    fmu2, fs22 = gp.predict(x_star, add_noise=False)
    if verbose:
        print(gp)
        print("optimised objective:", opt_result.fun)
        print("posterior mean range:", float(fmu.min()), float(fmu.max()))
        print("mean predictive sd before / after 20 more points:",
              float(np.sqrt(fs2).mean()), float(np.sqrt(fs22).mean()))
    return gp, fmu, fs2, fmu2, fs22


if __name__ == "__main__":
    main()

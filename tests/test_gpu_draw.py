"""GP.random_function against the reference's own seeded draws (tests/golden/draw_cases.npz, produced by gpyreg's
random_function, gaussian_process.py:2241-2329): posterior draws of three models and draws from the prior of a GP
without data, with and without observation noise.  The draw is T^T z + f_mu with T the Cholesky factor of the
predictive covariance (a device product here), so agreement is asked to 1e-7, not 1e-8."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _parse(name):
    tag, kname, mname, npar, N, D, data = str(name).split("|")
    degree, kernel = 0, kname
    if kname.startswith("matern"):
        kernel, degree = "matern", int(kname[6:])
    return tag, dict(kernel=kernel, degree=degree, mean=mname, noise=tuple(int(c) for c in npar)), int(N), int(D), data == "1"


def test_seeded_draws_match_the_reference():
    from test_gpu_api import _gp as mk

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "draw_cases.npz"), allow_pickle=False)
    for name in g["names"]:
        tag, model, N, D, data = _parse(name)
        gp = mk(model, D)
        if data:
            s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
            gp.update(X_new=g[tag + "_X"], y_new=g[tag + "_y"], s2_new=s2, hyp=g[tag + "_hyp"])
        else:
            gp.update(hyp=g[tag + "_hyp"])
            assert gp.y is None and np.size(gp.posteriors) == 3
        xs = g[tag + "_xs"]
        idx = int(tag[1:])
        for k in range(4):
            for key, noisy in (("_f", False), ("_y", True)):
                np.random.seed(900 + 10 * idx + k)
                mine = gp.random_function(xs, add_noise=noisy)
                ref = g[tag + key + str(k)]
                assert mine.shape == ref.shape == (7, 1)
                assert np.abs(mine - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (name, k, key)
        # a draw consumes the global generator exactly as the reference's does: the next number agrees
        np.random.seed(900 + 10 * idx)
        gp.random_function(xs, add_noise=True)
        after = np.random.standard_normal()
        np.random.seed(900 + 10 * idx)
        np.random.randint(0, 3)
        np.random.standard_normal((7, 1))
        np.random.standard_normal((7, 1))
        assert after == np.random.standard_normal()

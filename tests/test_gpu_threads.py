"""Two host threads, two GPs with different data on ONE device.  The library context (one stream, one workspace, one
resident copy of X and y) is shared: every entry takes the context's lock, and a GP method holds it from the upload
of ITS data to the call that uses them.  ctypes releases the GIL during a call, so without the lock the threads overlap
inside the library (and one GP would be evaluated on the other's data)."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(seed, N, D, matern):
    import gpyreg_amd as gpr

    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    cov = gpr.covariance_functions.Matern(5) if matern else gpr.covariance_functions.SquaredExponential()
    gp = gpr.GP(D, cov, gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True))
    hyp = np.concatenate([np.log(1.5) + 0.1 * rng.standard_normal((4, D)), 0.1 * rng.standard_normal((4, 1)),
                          np.log(0.1) + 0.1 * rng.standard_normal((4, 1)), 0.1 * rng.standard_normal((4, 1))], axis=1)
    gp.update(X_new=X, y_new=y, hyp=hyp)
    return gp, hyp, rng.uniform(-3, 3, (17, D))


def test_two_threads_two_gps_one_device():
    a, b = _model(1, 150, 2, True), _model(2, 260, 3, False)
    want = []
    for gp, hyp, xs in (a, b):
        want.append((gp.nll_batch(hyp, compute_grad=True), gp.predict(xs, separate_samples=True),
                     gp._GP__compute_nlZ(hyp[2], True, False)))
    errors = []

    def loop(k):
        gp, hyp, xs = (a, b)[k]
        (wn, wd), (wm, wv), (w1, wg) = want[k]
        try:
            for it in range(60):
                n, d = gp.nll_batch(hyp, compute_grad=True)
                m, v = gp.predict(xs, separate_samples=True)
                n1, g1 = gp._GP__compute_nlZ(hyp[2], True, False)
                if not (np.array_equal(n, wn) and np.array_equal(d, wd) and np.array_equal(m, wm)
                        and np.array_equal(v, wv) and n1 == w1 and np.array_equal(g1, wg)):
                    errors.append((k, it))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=loop, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads)
    assert not errors, errors

"""SliceSampler on its own (host only): a correlated 3-D Gaussian and a bounded, skewed density under the option
combinations that change the walk (step_out, adaptive, caller's widths, bounds, thinning, burn-in, a second call that
continues the chain), seeded, printed for a diff against the reference (protocol as in tools/api_sweep.py).  The
reference's log_prior / Metropolis / convergence diagnostics are out of scope: not exercised."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mod = importlib.import_module(os.environ.get("GPYREG_MODULE", "gpyreg_amd") + ".slice_sample")


def arr(v):
    a = np.asarray(v, dtype=float)
    return "%s sum %.12g absmax %.12g first %s" % (a.shape, a.sum(), np.abs(a).max(), " ".join("%.12g" % x for x in a.ravel()[:4]))


def main():
    A = np.array([[2.0, 0.6, -0.3], [0.6, 1.0, 0.2], [-0.3, 0.2, 0.5]])
    P = np.linalg.inv(A)
    gauss = lambda x: float(-0.5 * x @ P @ x)
    skew = lambda x: float(np.sum(2.0 * np.log(x) - 3.0 * x)) if np.all(x > 0) else -np.inf
    cases = [("gauss", gauss, np.zeros(3), None, None, None),
             ("gauss_w", gauss, np.ones(3), np.array([0.5, 2.0, 1.0]), None, None),
             ("gauss_b", gauss, np.array([0.5, 0.5, 0.5]), None, np.array([-1.0, -1.0, 0.0]), np.array([2.0, 1.5, 3.0])),
             ("skew", skew, np.ones(3), None, np.full(3, 1e-6), np.full(3, 20.0))]
    k = 0
    for name, f, x0, w, lb, ub in cases:
        for opts in ({}, {"step_out": True}, {"adaptive": False}, {"step_out": True, "adaptive": False}):
            k += 1
            np.random.seed(700 + k)
            o = dict(opts, display="off", diagnostics=False)
            try:
                s = mod.SliceSampler(f, x0.copy(), None if w is None else w.copy(), lb, ub, o)
                r1 = s.sample(40, thin=2, burn=10)
                r2 = s.sample(15)
                tag = "%s %s" % (name, "+".join(sorted(opts)) or "default")
                print(tag, "first", arr(r1["samples"]), "| f", arr(r1["f_vals"]))
                print(tag, "again", arr(r2["samples"]), "| f", arr(r2["f_vals"]))
                print(tag, "rng", "%.15g" % np.random.random_sample())
            except Exception as e:  # noqa: BLE001
                print(name, sorted(opts), "RAISES", type(e).__name__, str(e)[:90])
    for bad in (dict(thin=0), dict(thin=1, burn=-1)):
        try:
            mod.SliceSampler(gauss, np.zeros(3), None, None, None, {"display": "off", "diagnostics": False}).sample(5, **bad)
            print("bad", bad, "no error")
        except Exception as e:  # noqa: BLE001
            print("bad", sorted(bad.items()), "RAISES", type(e).__name__, str(e)[:70])
    try:
        mod.SliceSampler(lambda x: np.nan, np.zeros(3), None, None, None, {"display": "off", "diagnostics": False}).sample(5)
        print("nan start no error")
    except Exception as e:  # noqa: BLE001
        print("nan start RAISES", type(e).__name__, str(e)[:70])


if __name__ == "__main__":
    main()

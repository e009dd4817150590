"""Beyond N = 16384: the sizes the device library refused before round 6 (`gpc_max_n` was the reach of one 32-bit byte
offset over a k-major operand panel of the GEMM; gemm.h now advances a 64-bit base per k-slab) and the reference accepts
(it factorizes whatever fits host memory: gaussian_process.py:2415-2417, :2477-2484).

* N = 20480, D = 5, squared exponential: nlZ against the REFERENCE's own value and the gradient against the streamed
  oracle (oracle-derived and labelled so in tests/golden/big20k_case.npz, written only after its nlZ had been found equal
  to the reference's bit for bit): fp64 1e-8, fp32 1e-3 -- north_star's tolerances.
* N = 32768 (bench.py's `--config 6`), fp64, one and two samples: the size-independent properties -- a row of a batch
  equals its single evaluation bit for bit, NLL-only equals the NLL of NLL + gradient to rounding, the gradient agrees
  with a central difference along a random direction -- and a prediction against the posterior's own training targets.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _problem(N, S):
    import bench

    cfg = dict(bench.CONFIGS[6])
    try:
        bench.CONFIGS[6] = dict(cfg, N=N)
        return bench.synthetic_problem(6, S)
    finally:
        bench.CONFIGS[6] = cfg


def test_the_size_envelope_is_the_memory_budget():
    from gpyreg_amd import _lib

    lib = _lib.load()
    n64, n32 = lib.gpc_max_n(_lib.F64), lib.gpc_max_n(_lib.F32)
    assert n64 >= 32768 and n32 > n64 and n64 % 128 == 0  # (a 288 GB MI355X: 101 504 / 143 488)


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-8), ("f32", 1e-3)])
def test_n20480_against_the_reference_and_the_streamed_oracle(dtype, tol):
    import bench

    g = np.load(os.path.join(GOLDEN, "big20k_case.npz"), allow_pickle=False)
    N = int(g["N"])
    X, y, hyp = _problem(N, 2)
    assert np.array_equal(np.array([X.sum(), y.sum()]), g["Xsum"]) and np.array_equal(hyp[:1], g["hyp"])
    gp = bench.make_gp(6, dtype)
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp[:1], compute_grad=True)
    ref_n, ref_d = float(g["nlZ"][0]), g["dnlZ"][0]
    assert abs(nlz[0] - ref_n) <= tol * abs(ref_n), (nlz[0], ref_n)
    scale = np.maximum(np.abs(ref_d), np.abs(ref_d).max())
    assert (np.abs(dnlz[0] - ref_d) <= tol * scale).all(), (dnlz[0], ref_d)
    n0, _ = gp.nll_batch(hyp[:1], compute_grad=False)  # the N^3/3 plan (blocked solves) at this size
    assert abs(n0[0] - ref_n) <= tol * abs(ref_n)


def test_n32768_size_independent_properties():
    import bench

    N = bench.CONFIGS[6]["N"]
    X, y, hyp = _problem(N, 2)
    gp = bench.make_gp(6, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)  # S = 2
    assert np.isfinite(nlz).all() and np.isfinite(dnlz).all()
    for s in (0, 1):  # S = 1: a row of the batch carries the bits of its single evaluation
        n1, d1 = gp.nll_batch(hyp[s:s + 1], compute_grad=True)
        assert n1[0] == nlz[s] and np.array_equal(d1[0], dnlz[s]), s
    n0, _ = gp.nll_batch(hyp, compute_grad=False)
    assert np.allclose(n0, nlz, rtol=1e-12, atol=0)
    o0, _ = gp.nll_batch(hyp[1:2], compute_grad=False)
    assert o0[0] == n0[1]
    rng = np.random.default_rng(6)
    v = rng.standard_normal(hyp.shape[1])
    v /= np.linalg.norm(v)
    eps = 1e-4
    pm, _ = gp.nll_batch(np.stack([hyp[0] + eps * v, hyp[0] - eps * v]))
    num = (pm[0] - pm[1]) / (2 * eps)
    assert abs(num - dnlz[0] @ v) < 1e-6 * max(1.0, abs(num)), (num, dnlz[0] @ v)


def test_n32768_posterior_and_predict():
    """The posterior path at the same size (factor kept, K* product with a k-major operand of npad x mpad): the predictive
    mean at training inputs tracks the targets to within the fitted noise, the variance is positive and below the prior's,
    and one sample of a two-sample posterior set equals the one-sample set bit for bit."""
    import bench

    N = bench.CONFIGS[6]["N"]
    X, y, hyp = _problem(N, 2)
    gp = bench.make_gp(6, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp)
    xs = np.concatenate([X[:200], X[-200:]])
    mu, s2 = gp.predict(xs, separate_samples=True, add_noise=False)
    assert np.isfinite(mu).all() and (s2 > 0).all()
    sf2 = np.exp(2 * hyp[:, 5])
    assert (s2 < sf2[None, :]).all()
    resid = mu[:, 0] - np.concatenate([y[:200, 0], y[-200:, 0]])
    assert np.sqrt(np.mean(resid ** 2)) < 3 * np.exp(hyp[0, 6])  # noise std of sample 0
    gp1 = bench.make_gp(6, "f64")
    gp1.update(X_new=X, y_new=y, hyp=hyp[1:2])
    mu1, s21 = gp1.predict(xs, separate_samples=True, add_noise=False)
    assert np.array_equal(mu1[:, 0], mu[:, 1]) and np.array_equal(s21[:, 0], s2[:, 1])


def test_n65536_more_than_2_31_elements_per_matrix():
    """N = 65536: every matrix of a sample holds 2^32 elements (34 GB in fp64, three of them per sample) -- beyond a 32-bit
    ELEMENT index, not only a 32-bit byte offset.  One sample: NLL-only (the N^3/3 plan with blocked solves) equals the NLL
    of NLL + gradient to rounding, and the gradient agrees with a central difference along a random direction."""
    import bench
    from gpyreg_amd import _lib

    N = 65536
    if _lib.load().gpc_max_n(_lib.F64) < N:
        pytest.skip("this device's memory does not hold three 34 GB slabs")
    X, y, hyp = _problem(N, 1)
    gp = bench.make_gp(6, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    assert np.isfinite(nlz).all() and np.isfinite(dnlz).all()
    rng = np.random.default_rng(65536)
    v = rng.standard_normal(hyp.shape[1])
    v /= np.linalg.norm(v)
    eps = 1e-4
    trio, _ = gp.nll_batch(np.stack([hyp[0], hyp[0] + eps * v, hyp[0] - eps * v]), compute_grad=False)
    assert abs(trio[0] - nlz[0]) <= 1e-11 * abs(nlz[0]), (trio[0], nlz[0])
    num = (trio[1] - trio[2]) / (2 * eps)
    assert abs(num - dnlz[0] @ v) < 1e-6 * max(1.0, abs(num)), (num, dnlz[0] @ v)


def test_n32768_fp32_properties():
    """The fp32 mode beyond its former ceiling (23 168): N = 32 768, one and two samples -- batch row == single evaluation bit for
    bit, NLL-only == NLL of NLL + gradient at fp32 accuracy, the fp32 NLL within 1e-3 of the fp64 device value (itself held
    to the properties above), the gradient within 1e-3 per component of the fp64 device gradient."""
    import bench

    N = bench.CONFIGS[6]["N"]
    X, y, hyp = _problem(N, 2)
    gp32 = bench.make_gp(6, "f32")
    gp32.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    n32, d32 = gp32.nll_batch(hyp, compute_grad=True)
    one, done = gp32.nll_batch(hyp[1:2], compute_grad=True)
    assert one[0] == n32[1] and np.array_equal(done[0], d32[1])
    m32, _ = gp32.nll_batch(hyp, compute_grad=False)
    # (two blockings of an fp32 factorization -- the N^3/3 plan solves against the factor, the gradient plan multiplies by
    # inverses -- agree to what fp32 supports on the quadratic form, cond x eps32; each is held to 1e-3 of fp64 below)
    assert np.allclose(m32, n32, rtol=5e-4, atol=0), (m32, n32)
    gp64 = bench.make_gp(6, "f64")
    gp64.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    n64, d64 = gp64.nll_batch(hyp, compute_grad=True)
    assert np.abs(n32 - n64).max() <= 1e-3 * np.abs(n64).max(), (n32, n64)
    assert np.abs(m32 - n64).max() <= 1e-3 * np.abs(n64).max(), (m32, n64)
    assert (np.abs(d32 - d64) / np.maximum(np.abs(d64), np.abs(d64).max(1, keepdims=True))).max() <= 1e-3


def test_beyond_the_envelope_is_refused_before_anything_is_allocated():
    """N above gpc_max_n: the batch entry points answer -2 with a message about the device's memory (no allocation is
    attempted, the context stays usable) -- the one size limit left, and it is the device's, not an addressing mode's."""
    import gpyreg_amd as gpr
    from gpyreg_amd import _lib

    n_max = _lib.load().gpc_max_n(_lib.F64)
    N, D = n_max + 128, 2
    rng = np.random.default_rng(1)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True))
    hyp = np.array([[0.5, 0.5, 0.0, np.log(0.1), 0.0]])
    gp = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    with pytest.raises(RuntimeError) as e:
        gp.nll_batch(hyp, compute_grad=False)
    assert "memory" in str(e.value) and "gpc_max_n" in str(e.value)
    small = gpr.GP(D, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                   gpr.noise_functions.GaussianNoise(constant_add=True))
    small.update(X_new=X[:300], y_new=y[:300], hyp=hyp, compute_posterior=False)
    assert np.isfinite(small.nll_batch(hyp, compute_grad=True)[0]).all()


def test_the_product_library_does_not_know_the_rejected_schedules():
    from gpyreg_amd import _lib

    if _lib.is_experiments_build():
        pytest.skip("the experiments build is loaded")
    ctx = _lib.context(0)
    assert ctx.get_option("experiments") == 0
    for name in ("dag", "indep", "rect_min", "rl_panel", "dag_timeout_ms"):
        with pytest.raises(RuntimeError) as e:
            ctx.set_option(name, 1)
        assert "unknown option" in str(e.value)

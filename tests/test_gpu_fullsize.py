"""Full-size GPU checks at BASELINE.json's configurations: nlZ and gradient against values
produced by the reference itself at full size (tests/golden/fullsize_cases.npz), in fp64
(1e-8) and fp32 (1e-3), plus size-independent properties."""

import numpy as np
import pytest

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _gp(cfg):
    import bench

    return bench.make_gp(cfg, "f64")


def test_cfg2_matches_survey_reference_value():
    import bench

    X, y, hyp = bench.synthetic_problem(2, 1)
    # bench's generator and the oracle's agree
    _, X2, y2, hyp2 = orc.synthetic_problem(2)
    assert np.array_equal(X, X2) and np.array_equal(y, y2) and np.array_equal(hyp, hyp2)
    gp = _gp(2)
    gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    assert abs(nlz[0] - (-1166.298896135079)) < 1e-8 * 1166.3


def test_cfg3_reference_value_gradient_and_batch_consistency():
    import bench

    X, y, hyp = bench.synthetic_problem(3, 16)
    gp = _gp(3)
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    assert abs(nlz[0] - 486.629241506640) < 1e-8 * 486.63  # reference value, SURVEY 8(d)
    assert abs(nlz[1] - 702.291395297840) < 1e-8 * 702.29
    assert np.isfinite(dnlz).all()
    # batch of 16 == one at a time (no cross-talk between samples), NLL-only == with grad
    n1, d1 = gp.nll_batch(hyp[5:6], compute_grad=True)
    assert n1[0] == nlz[5] and np.array_equal(d1[0], dnlz[5])
    n0, _ = gp.nll_batch(hyp[:4], compute_grad=False)
    assert np.allclose(n0, nlz[:4], rtol=1e-13)
    # directional derivative: central difference along a random direction
    rng = np.random.default_rng(0)
    v = rng.standard_normal(hyp.shape[1])
    v /= np.linalg.norm(v)
    eps = 1e-5
    pm, _ = gp.nll_batch(np.stack([hyp[0] + eps * v, hyp[0] - eps * v]))
    num = (pm[0] - pm[1]) / (2 * eps)
    assert abs(num - dnlz[0] @ v) < 1e-6 * max(1.0, abs(num))


def test_headline_sizes_match_reference_goldens_1e8():
    """north_star bar at the sizes it is quoted on: nlZ AND the full gradient of cfg2 (N=2048,
    sample 0) and cfg3 (N=4096, samples 0, 1, 15 of the 16) against values produced by running
    the reference itself at full size (tests/golden/fullsize_cases.npz, make_golden.py fullsize),
    1e-8 relative; the inputs are regenerated from the seed and checked against the fixture."""
    import os

    import bench

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_cases.npz"), allow_pickle=False)
    for cfg, S in [(2, 1), (3, 16)]:
        X, y, hyp = bench.synthetic_problem(cfg, S)
        rows = g[f"cfg{cfg}_rows"]
        assert np.array_equal(hyp[rows], g[f"cfg{cfg}_hyp"])
        assert np.allclose([X.sum(), y.sum()], g[f"cfg{cfg}_Xsum"], rtol=1e-13)
        gp = _gp(cfg)
        gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
        for k, s in enumerate(rows):
            rn, rd = g[f"cfg{cfg}_nlZ"][k], g[f"cfg{cfg}_dnlZ"][k]
            e_n = abs(nlz[s] - rn) / max(1.0, abs(rn))
            e_d = (np.abs(dnlz[s] - rd) / np.maximum(np.abs(rd), np.abs(rd).max())).max()
            print(f"cfg{cfg} sample {s}: nlZ rel err {e_n:.2e}, gradient rel err {e_d:.2e}")
            assert e_n < 1e-8 and e_d < 1e-8, (cfg, s, e_n, e_d)


def test_fp32_full_size_against_reference_goldens():
    """fp32 mode at N=2048 / N=4096 against the same reference values: 1e-3 relative."""
    import os

    import bench

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_cases.npz"), allow_pickle=False)
    for cfg, S in [(2, 1), (3, 16)]:
        X, y, hyp = bench.synthetic_problem(cfg, S)
        rows = g[f"cfg{cfg}_rows"]
        gp = bench.make_gp(cfg, "f32")
        gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
        nlz, dnlz = gp.nll_batch(hyp[rows], compute_grad=True)
        for k in range(len(rows)):
            rn, rd = g[f"cfg{cfg}_nlZ"][k], g[f"cfg{cfg}_dnlZ"][k]
            e_n = abs(nlz[k] - rn) / max(1.0, abs(rn))
            e_d = (np.abs(dnlz[k] - rd) / np.maximum(np.abs(rd), np.abs(rd).max())).max()
            print(f"fp32 cfg{cfg} sample {rows[k]}: nlZ rel err {e_n:.2e}, gradient rel err {e_d:.2e}")
            assert e_n < 1e-3 and e_d < 1e-3, (cfg, k, e_n, e_d)


def test_predict_interpolates_at_full_size():
    import bench

    X, y, hyp = bench.synthetic_problem(2, 2)
    gp = _gp(2)
    gp.update(X_new=X, y_new=y, hyp=hyp)
    mu, s2 = gp.predict(X[:300], separate_samples=True)
    # posterior variance at training inputs is below the prior variance and non-negative,
    # the mean tracks the targets to within a few noise standard deviations
    sf2 = np.exp(2 * hyp[:, 5])
    assert (s2 >= 0).all() and (s2 < sf2[None, :]).all()
    assert np.abs(mu - y[:300]).max() < 1.0
    # alpha reproduces the data equation  (K + sn2 I) alpha = y - m  through predict:
    # mean at training points = y - sn2 * alpha
    a = gp.posteriors[0].alpha[:300, 0]
    sn2 = np.exp(2 * hyp[0, 6])
    assert np.allclose(mu[:, 0], y[:300, 0] - sn2 * a, rtol=1e-7, atol=1e-8)


def test_fp32_mode_within_1e3_of_fp64():
    """north_star: fp32 factorization within 1e-3 relative of the fp64 result."""
    import bench

    for cfg, N in [(4, 2048), (3, 1024)]:
        bench_cfg = dict(bench.CONFIGS[cfg])
        try:
            bench.CONFIGS[cfg] = dict(bench_cfg, N=N)
            X, y, hyp = bench.synthetic_problem(cfg, 3)
            res = {}
            for dt in ("f64", "f32"):
                gp = bench.make_gp(cfg, dt)
                gp.update(X_new=X, y_new=y, hyp=hyp)
                res[dt] = gp.nll_batch(hyp, compute_grad=True) + gp.predict(X[:50], separate_samples=True)
        finally:
            bench.CONFIGS[cfg] = bench_cfg
        n64, d64, mu64, s64 = res["f64"]
        n32, d32, mu32, s32 = res["f32"]
        assert np.abs(n32 - n64).max() <= 1e-3 * np.abs(n64).max()
        assert (np.abs(d32 - d64) / np.maximum(np.abs(d64), np.abs(d64).max(1, keepdims=True))).max() <= 1e-3
        assert np.abs(mu32 - mu64).max() <= 1e-3 * max(1.0, np.abs(mu64).max())
        assert np.abs(s32 - s64).max() <= 1e-3


def test_cfg5_and_cfg4_sizes_size_independent_properties():
    """N = 8192 (fp64) and N = 16384 (fp32): batch == single, NLL-only == NLL of NLL+grad,
    directional finite difference of the gradient."""
    import bench

    for cfg, S, dtype, rtol in [(5, 3, "f64", 1e-6), (4, 1, "f32", 5e-2)]:
        X, y, hyp = bench.synthetic_problem(cfg, max(S, 2))
        gp = bench.make_gp(cfg, dtype)
        gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
        nlz, dnlz = gp.nll_batch(hyp[:S], compute_grad=True)
        assert np.isfinite(nlz).all() and np.isfinite(dnlz).all()
        n0, _ = gp.nll_batch(hyp[:S], compute_grad=False)
        assert np.allclose(n0, nlz, rtol=1e-12 if dtype == "f64" else 1e-5)
        if S > 1:
            n1, d1 = gp.nll_batch(hyp[1:2], compute_grad=True)
            assert n1[0] == nlz[1] and np.array_equal(d1[0], dnlz[1])
        rng = np.random.default_rng(cfg)
        v = rng.standard_normal(hyp.shape[1])
        v /= np.linalg.norm(v)
        eps = 1e-4 if dtype == "f64" else 1e-2
        pm, _ = gp.nll_batch(np.stack([hyp[0] + eps * v, hyp[0] - eps * v]))
        num = (pm[0] - pm[1]) / (2 * eps)
        assert abs(num - dnlz[0] @ v) < rtol * max(1.0, abs(num))


def test_uneven_sample_queues_match_single_evaluations():
    """13 samples at N=2048: the persistent GEMM serves 8 XCD queues of 2,2,2,2,2,1,1,1 samples (and
    steals across them); every sample must equal its own single evaluation bit for bit (which runs
    through the non-persistent 64-tile kernels), with and without the gradient."""
    import bench

    X, y, hyp = bench.synthetic_problem(2, 13)
    gp = _gp(2)
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    n0, _ = gp.nll_batch(hyp, compute_grad=False)
    assert np.isfinite(nlz).all() and np.isfinite(dnlz).all()
    assert np.allclose(n0, nlz, rtol=1e-13)
    for s in (0, 4, 7, 8, 12):
        n1, d1 = gp.nll_batch(hyp[s:s + 1], compute_grad=True)
        assert n1[0] == nlz[s] and np.array_equal(d1[0], dnlz[s])


@pytest.mark.parametrize("kernel,degree,noise", [("rq", 0, (1, 0, 0)), ("se_iso", 0, (1, 1, 0)), ("matern_iso", 3, (1, 2, 0)),
                                                 ("matern", 3, (1, 0, 1)), ("matern", 1, (1, 0, 0))])
def test_mid_size_every_kernel_family_against_the_oracle(kernel, degree, noise):
    """N = 1408 (11 tiles: odd splits at every level of the recursion), 10 samples so that the large
    products run as persistent launches with uneven XCD queues: NLL and gradient of the kernel
    families and noise models the BASELINE configs do not touch, against the CPU oracle (1e-8)."""
    from test_gpu_api import _gp as mk

    N, D, S = 1408, 3, 10
    rng = np.random.default_rng(N + degree + len(kernel))
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
    s2 = 0.01 * rng.uniform(0.5, 2.0, (N, 1)) if noise[1] else None
    model = dict(kernel=kernel, degree=degree, mean="const", noise=noise)
    cov_N = orc.cov_count(kernel, D)
    noise_N = orc.noise_count(noise)
    nl = D if cov_N > 2 else 1
    base = np.concatenate([np.log(1.5 * np.sqrt(D)) * np.ones(nl), np.zeros(cov_N - nl),
                           [np.log(0.1)] + [0.0] * (noise_N - 1), [0.0]])
    hyp = base + 0.1 * rng.standard_normal((S, base.size))
    gp = mk(model, D)
    gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp[:1], compute_posterior=False)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    gp32 = mk(model, D, dtype="f32")  # the fp32 path against the SAME oracle values, 1e-3
    gp32.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp[:1], compute_posterior=False)
    nlz32, dnlz32 = gp32.nll_batch(hyp, compute_grad=True)
    for s in (0, 9):
        rn, rd = orc.core(model, hyp[s], X, y, s2, 1, 1)
        assert abs(nlz[s] - rn) <= 1e-8 * max(1.0, abs(rn)), (kernel, s)
        assert abs(nlz32[s] - rn) <= 1e-3 * max(1.0, abs(rn)), (kernel, s, "f32")
        if kernel == "matern" and degree == 1:  # NaN length-scale gradients on purpose
            assert np.array_equal(np.isnan(dnlz[s]), np.isnan(rd))
            assert np.array_equal(np.isnan(dnlz32[s]), np.isnan(rd))
            m = ~np.isnan(rd)
        else:
            m = np.ones(rd.shape, bool)
        scale = np.maximum(np.abs(rd[m]), np.abs(rd[m]).max())
        assert (np.abs(dnlz[s][m] - rd[m]) <= 1e-8 * scale).all(), (kernel, s)
        assert (np.abs(dnlz32[s][m] - rd[m]) <= 1e-3 * scale).all(), (kernel, s, "f32")


def test_deferred_inverse_products_do_not_change_a_bit():
    """plan.h's deferred U = T21 W11 (side stream, CU-reserving persistent launch) reorders launches, not
    arithmetic: NLL, gradient and posteriors with the schedule forced on (every node >= 512), automatic and
    off must be identical bit for bit (N = 2304: odd splits; S = 6 and 16)."""
    import bench
    from gpyreg_amd import _lib

    ctx = _lib.context(0)
    bench_cfg = dict(bench.CONFIGS[3])
    try:
        bench.CONFIGS[3] = dict(bench_cfg, N=2304)
        for S in (6, 16):
            X, y, hyp = bench.synthetic_problem(3, S)
            xs = X[:33] + 0.01
            res = []
            for dmin in (0, -1, 512):
                ctx.set_option("defer_min", dmin)
                gp = bench.make_gp(3, "f64")
                gp.update(X_new=X, y_new=y, hyp=hyp)
                res.append(gp.nll_batch(hyp, compute_grad=True) + gp.predict(xs, separate_samples=True))
            for r in res[1:]:
                for a, b in zip(res[0], r):
                    assert np.array_equal(a, b)
    finally:
        bench.CONFIGS[3] = bench_cfg
        ctx.set_option("defer_min", -1)


def test_dual_launches_do_not_change_a_bit():
    """plan.h launches the syrk A22 -= T21 T21^T and the inverse product U = T21 W11 of a node as ONE grid where
    both are small launches: a launch less per node, the same arithmetic.  NLL, gradient and posteriors with the
    dual launches off and on must be identical bit for bit (N = 700 and 2304, S = 1 and 6)."""
    import bench
    from gpyreg_amd import _lib

    ctx = _lib.context(0)
    bench_cfg = dict(bench.CONFIGS[3])
    try:
        for N, S in ((700, 1), (2304, 6)):
            bench.CONFIGS[3] = dict(bench_cfg, N=N)
            X, y, hyp = bench.synthetic_problem(3, S)
            xs = X[:33] + 0.01
            res = []
            for dual in (0, 1):
                ctx.set_option("dual_launch", dual)
                gp = bench.make_gp(3, "f64")
                gp.update(X_new=X, y_new=y, hyp=hyp)
                res.append(gp.nll_batch(hyp, compute_grad=True) + gp.predict(xs, separate_samples=True))
            for a, b in zip(res[0], res[1]):
                assert np.array_equal(a, b)
    finally:
        bench.CONFIGS[3] = bench_cfg
        ctx.set_option("dual_launch", 1)


def test_pipelined_leaf_does_not_change_a_bit_of_the_pipeline():
    """The whole path -- NLL, gradient, posteriors, predictions -- with the pipelined leaf (default) and with the
    phase-ordered one it replaced: identical bit for bit (N = 700 and 2304, incl. a padded last leaf; fp64 and fp32)."""
    import bench
    from gpyreg_amd import _lib

    ctx = _lib.context(0)
    bench_cfg = dict(bench.CONFIGS[3])
    try:
        for N, S, dtype in ((700, 3, "f64"), (2304, 5, "f64"), (700, 3, "f32")):
            bench.CONFIGS[3] = dict(bench_cfg, N=N)
            X, y, hyp = bench.synthetic_problem(3, S)
            xs = X[:33] + 0.01
            res = []
            for leaf in (3, 5):
                ctx.set_option("leaf", leaf)
                gp = bench.make_gp(3, dtype)
                gp.update(X_new=X, y_new=y, hyp=hyp)
                res.append(gp.nll_batch(hyp, compute_grad=True) + gp.predict(xs, separate_samples=True))
            for a, b in zip(res[0], res[1]):
                assert np.array_equal(a, b)
    finally:
        bench.CONFIGS[3] = bench_cfg
        ctx.set_option("leaf", 5)

"""GPU edge cases: wide inputs (D > 32: chunked LDS staging), sample chunking under a
small memory budget, fp32 posteriors, degenerate sizes."""

import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(N, D, S, kernel="se", degree=0, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
    cov_N = orc.cov_count(kernel, D)
    base = np.concatenate([np.log(1.5 * np.sqrt(D)) * np.ones(D if cov_N > 2 else 1), np.zeros(cov_N - (D if cov_N > 2 else 1)),
                           [np.log(0.1), 0.0]])
    hyp = base + 0.1 * rng.standard_normal((S, base.size))
    model = dict(kernel=kernel, degree=degree, mean="const", noise=(1, 0, 0))
    return model, X, y, hyp


def _gp(model, D, dtype="f64"):
    from test_gpu_api import _gp as mk

    return mk(model, D, dtype)


@pytest.mark.parametrize("kernel,degree,D", [("se", 0, 40), ("matern", 5, 70), ("rq", 0, 33)])
def test_wide_inputs_chunked_staging(kernel, degree, D):
    model, X, y, hyp = _problem(150, D, 2, kernel, degree, seed=D)
    gp = _gp(model, D)
    gp.update(X_new=X, y_new=y, hyp=hyp)
    nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
    K, dK = gp.covariance.compute(hyp[0, :orc.cov_count(kernel, D)], X[:20], compute_grad=True)
    rK, rdK = orc.covariance(kernel, hyp[0, :orc.cov_count(kernel, D)], X[:20], compute_grad=True, degree=degree)
    assert np.allclose(K, rK, rtol=1e-12) and np.allclose(dK, rdK, rtol=1e-10, atol=1e-13)
    for s in range(2):
        rn, rd = orc.core(model, hyp[s], X, y, None, 1, 1)
        assert abs(nlz[s] - rn) <= 1e-8 * max(1.0, abs(rn))
        assert np.abs(dnlz[s] - rd).max() <= 1e-8 * np.abs(rd).max()
    mu, s2 = gp.predict(X[:9], separate_samples=True)
    rmu, rs2 = orc.predict(model, orc.posteriors(model, hyp, X, y, None), X, y, X[:9], separate_samples=True)
    assert np.allclose(mu, rmu, atol=1e-8) and np.allclose(s2, rs2, atol=1e-8)


def test_tiny_and_single_point():
    for N in (1, 2, 5):
        model, X, y, hyp = _problem(N, 2, 2, seed=N)
        gp = _gp(model, 2)
        gp.update(X_new=X, y_new=y, hyp=hyp)
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
        for s in range(2):
            rn, rd = orc.core(model, hyp[s], X, y, None, 1, 1)
            assert abs(nlz[s] - rn) <= 1e-10 * max(1.0, abs(rn)) and np.allclose(dnlz[s], rd, rtol=1e-9, atol=1e-12)
        mu, s2 = gp.predict(np.zeros((1, 2)))
        assert mu.shape == (1, 1) and s2.shape == (1, 1) and s2[0, 0] >= 0


def test_fp32_posterior_fetch_and_fields():
    model, X, y, hyp = _problem(200, 3, 2, seed=3)
    g64, g32 = _gp(model, 3, "f64"), _gp(model, 3, "f32")
    g64.update(X_new=X, y_new=y, hyp=hyp)
    g32.update(X_new=X, y_new=y, hyp=hyp)
    for p64, p32 in zip(g64.posteriors, g32.posteriors):
        assert np.abs(p32.L - p64.L).max() <= 1e-4 * np.abs(p64.L).max()
        assert np.abs(p32.alpha - p64.alpha).max() <= 1e-3 * np.abs(p64.alpha).max()
        assert np.array_equal(p32.sW, p64.sW) and p32.L_chol == p64.L_chol


def test_sample_chunking_under_small_memory_budget():
    """S=5 samples with a budget that fits 2 at a time must equal the unchunked result."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_edge import _problem, _gp
model, X, y, hyp = _problem(300, 3, 5, "matern", 5, seed=11)
gp = _gp(model, 3)
gp.update(X_new=X, y_new=y, hyp=hyp)
nlz, dnlz = gp.nll_batch(hyp, compute_grad=True)
mu, s2 = gp.predict(X[:17], separate_samples=True)
_, C = gp.predict_full(X[:17])
np.savez(sys.argv[1], nlz=nlz, dnlz=dnlz, mu=mu, s2=s2, C=C, a=np.stack([p.alpha for p in gp.posteriors]))
""" % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile

    outs = []
    for budget in (None, "6"):  # 384^2*8*3 = 3.5 MB per sample -> 80% of 6 MB holds one
        env = dict(os.environ)
        if budget:
            env["GPC_MEM_BUDGET_MB"] = budget
        f = tempfile.NamedTemporaryFile(suffix=".npz", delete=False).name
        subprocess.run([sys.executable, "-c", code, f], check=True, env=env, cwd=ROOT, timeout=300)
        outs.append(np.load(f))
    for k in ("nlz", "dnlz", "mu", "s2", "C", "a"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_replayed_launch_graph_follows_new_inputs_and_new_data():
    """Small problems replay a cached hipGraph of the launch sequence: the replays must see the
    hyperparameters of each call (device buffers are refilled, pointers are not), a changed data
    set of the same shape, a changed shape, and a jitter retry (different sample count)."""
    model, X, y, hyp = _problem(200, 3, 4, kernel="matern", degree=3, seed=5)
    gp = _gp(model, 3)
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    ref = [orc.core(model, hyp[s], X, y, None, 1, 1) for s in range(4)]
    for rep in range(3):  # same shape again and again, different hyperparameters each call
        for s in (0, 1, 2, 3, 1):
            n, d = gp._GP__compute_nlZ(hyp[s], True, False)
            assert abs(n - ref[s][0]) <= 1e-9 * max(1.0, abs(ref[s][0]))
            assert np.allclose(d, ref[s][1], rtol=1e-8, atol=1e-10)
    nb, db = gp.nll_batch(hyp, compute_grad=True)  # another sample count: another graph
    assert np.allclose(nb, [r[0] for r in ref], rtol=1e-9)
    # new data of the same shape through the same graph
    model2, X2, y2, _ = _problem(200, 3, 4, kernel="matern", degree=3, seed=6)
    gp2 = _gp(model, 3)
    gp2.update(X_new=X2, y_new=y2, hyp=hyp[:1], compute_posterior=False)
    n2, d2 = gp2._GP__compute_nlZ(hyp[0], True, False)
    r2 = orc.core(model, hyp[0], X2, y2, None, 1, 1)
    assert abs(n2 - r2[0]) <= 1e-9 * max(1.0, abs(r2[0])) and np.allclose(d2, r2[1], rtol=1e-8, atol=1e-10)
    # and the first problem is still right afterwards (its graph, if still cached, reads X from dX)
    gp3 = _gp(model, 3)
    gp3.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    n3, _ = gp3._GP__compute_nlZ(hyp[2], True, False)
    assert abs(n3 - ref[2][0]) <= 1e-9 * max(1.0, abs(ref[2][0]))

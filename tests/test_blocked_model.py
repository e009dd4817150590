"""CPU check of the device launch plan (tests/blocked_model.py) against NumPy."""

import numpy as np
import pytest

import blocked_model as bm


def _spd(n, rng):
    X = rng.uniform(-3, 3, (n, 2))
    d = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d) / 0.01 + np.eye(n)


@pytest.mark.parametrize("n,tile", [(4, 4), (8, 4), (12, 4), (20, 4), (28, 4), (37, 4), (64, 8)])
@pytest.mark.parametrize("post_mode", [False, True])
def test_plan_factor_inverse(n, tile, post_mode):
    rng = np.random.default_rng(n * 10 + tile)
    A0 = _spd(n, rng)
    A = bm.pad_identity(A0, tile)
    npad = A.shape[0]
    Aref = A.copy()
    A[np.triu_indices(npad, 1)] = np.nan  # the plan must never read the upper part of A
    W = np.zeros((npad, npad))
    T = np.full((npad, npad), np.nan)
    log = {}
    info = bm.potrf_inv(A, W, T, 0, npad, tile, True, post_mode, log)
    assert info == 0
    Lref = np.linalg.cholesky(Aref)
    Wref = np.linalg.inv(Lref)
    assert np.allclose(np.diag(A), np.diag(Lref), rtol=1e-12)
    if post_mode:
        assert np.allclose(np.tril(A), Lref, rtol=1e-10, atol=1e-12)
    assert np.allclose(W, Wref, rtol=1e-9, atol=1e-12)
    assert np.all(W[np.triu_indices(npad, 1)] == 0)
    Kinv = np.full((npad, npad), np.nan)
    bm.lauum(Kinv, W, npad, tile, log)
    ref = np.linalg.inv(Aref)
    il = np.tril_indices(npad)
    assert np.allclose(Kinv[il], ref[il], rtol=1e-8, atol=1e-12)
    # identity padding stays exactly inert
    if npad > n:
        assert np.all(Kinv[n:, :n] == 0)
        assert np.array_equal(np.tril(Kinv[n:, n:]), np.eye(npad - n))
    # flop count approaches n^3 (potrf+trtri+lauum) for many tiles
    if npad // tile >= 8:
        assert log["flops"] < 1.6 * npad**3


def test_plan_nll_only_mode_skips_top_level_inverse():
    rng = np.random.default_rng(5)
    n, tile = 32, 4
    A0 = _spd(n, rng)
    A = A0.copy()
    W = np.zeros((n, n))
    T = np.full((n, n), np.nan)
    log = {}
    assert bm.potrf_inv(A, W, T, 0, n, tile, False, False, log) == 0
    Lref = np.linalg.cholesky(A0)
    assert np.allclose(np.diag(A), np.diag(Lref), rtol=1e-12)
    r = rng.standard_normal(n)
    z = r.copy()
    bm.forward_solve(T, W, z, 0, n, tile, False)
    assert np.allclose(z, np.linalg.solve(Lref, r), rtol=1e-9, atol=1e-12)
    full = {}
    A2, W2, T2 = A0.copy(), np.zeros((n, n)), np.full((n, n), np.nan)
    bm.potrf_inv(A2, W2, T2, 0, n, tile, True, False, full)
    assert log["flops"] < 0.75 * full["flops"]


def test_plan_reports_first_bad_pivot():
    n, tile = 16, 4
    A = np.eye(n)
    A[9, 9] = -1.0
    W = np.zeros((n, n))
    T = np.zeros((n, n))
    assert bm.potrf_inv(A, W, T, 0, n, tile, True, False) == 10


@pytest.mark.parametrize("n,tile", [(12, 4), (37, 4), (64, 8)])
def test_stable_mode_plan_is_the_same_factorization(n, tile):
    """plan.h's stable mode (jitter retries): the refinement launches T21 += (A21 - T21 L11^T) W11^T read L11 out
    of A -- complete there because L21 is kept at every node, with clean diagonal tiles -- and change nothing but
    rounding on a well-conditioned matrix."""
    rng = np.random.default_rng(n + tile)
    A0 = _spd(n, rng)
    out = []
    for stable in (False, True):
        A = bm.pad_identity(A0, tile)
        npad = A.shape[0]
        A[np.triu_indices(npad, 1)] = np.nan
        W = np.zeros((npad, npad))
        T = np.full((npad, npad), np.nan)
        assert bm.potrf_inv(A, W, T, 0, npad, tile, True, False, stable=stable,
                            leaf_fn=lambda a, w: bm.leaf_panels(a, w, 2, refine=stable)) == 0
        out.append((np.tril(A) if stable else None, W))
    L = np.linalg.cholesky(bm.pad_identity(A0, tile))
    assert np.allclose(out[1][0], L, rtol=1e-10, atol=1e-12)
    assert np.allclose(out[0][1], out[1][1], rtol=1e-9, atol=1e-12)


def test_stable_mode_needs_no_more_jitter_than_lapack_on_a_singular_matrix():
    """The reason stable mode exists (tests/analysis/jitter_model.py): near-duplicate inputs, tiny noise.  The fast plan
    (explicit-inverse panel solves) fails at jitter levels where LAPACK succeeds; the stable plan does not."""
    import scipy.linalg as sla

    rng = np.random.default_rng(3)
    worse = {False: 0, True: 0}
    for trial in range(12):
        N = 40 + 3 * trial
        X = rng.uniform(-3, 3, (N, 2))
        X[N // 2:] = X[:N - N // 2] + 1e-7 * rng.standard_normal((N - N // 2, 2))
        K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
        s = 10.0 ** rng.uniform(-20, -16)

        def first_level(fact):
            for k in range(14):
                with np.errstate(all="ignore"):
                    if fact(K + 10.0 ** k * s * np.eye(N)):
                        return k
            return 99

        def lapack(A):
            try:
                sla.cholesky(A, lower=True, check_finite=False)
                return True
            except sla.LinAlgError:
                return False

        def plan(stable):
            def f(A):
                P = bm.pad_identity(A, 32)
                n = P.shape[0]
                return bm.potrf_inv(P, np.zeros((n, n)), np.zeros((n, n)), 0, n, 32, True, False, stable=stable,
                                    leaf_fn=lambda a, w: bm.leaf_panels(a, w, 16, refine=stable)) == 0
            return f

        base = first_level(lapack)
        for stable in (False, True):
            worse[stable] += first_level(plan(stable)) > base
    assert worse[False] >= 6, worse   # the fast plan is one-sidedly worse ...
    assert worse[True] <= 2, worse    # ... the stable one is not


@pytest.mark.parametrize("n,tile,blk", [(32, 4, 4), (32, 4, 8), (44, 4, 8), (72, 4, 16), (64, 8, 8), (37, 4, 12)])
def test_plan_nll_only_blocked_triangular_solves(n, tile, blk):
    """plan.h: potrf_nll / trsm_nll / forward_solve_nll -- only diagonal blocks of at most `blk` rows are inverted,
    the panels above are blocked solves against the factor in the scratch; N^3/3 flops; the strictly upper part of A
    and everything outside the written regions is never read (NaN poison)."""
    rng = np.random.default_rng(n + tile + blk)
    A0 = _spd(n, rng)
    A = bm.pad_identity(A0, tile)
    npad = A.shape[0]
    Aref = A.copy()
    A[np.triu_indices(npad, 1)] = np.nan
    W = np.full((npad, npad), np.nan)
    for o in range(0, npad, tile):  # leaves write whole diagonal tiles of W
        W[o:o + tile, o:o + tile] = 0
    T = np.full((npad, npad), np.nan)
    log = {}
    assert bm.potrf_nll(A, W, T, 0, npad, tile, blk, log) == 0
    Lref = np.linalg.cholesky(Aref)
    assert np.allclose(np.diag(A), np.diag(Lref), rtol=1e-12)
    r = rng.standard_normal(npad)
    z = r.copy()
    bm.forward_solve_nll(T, W, z, 0, npad, tile, blk)
    assert np.allclose(z, np.linalg.solve(Lref, r), rtol=1e-9, atol=1e-12)
    if npad // tile >= 8 and blk <= 2 * tile:
        old = {}
        A2 = bm.pad_identity(A0, tile)
        bm.potrf_inv(A2, np.zeros((npad, npad)), np.zeros((npad, npad)), 0, npad, tile, False, False, old)
        assert log["flops"] < 0.97 * old["flops"]            # fewer flops than inverting every left child ...
        assert log["flops"] < 1.45 * npad**3 / 3               # ... and close to N^3/3 (+ the small inverses)


@pytest.mark.parametrize("n,tile,panel", [(32, 4, 8), (44, 4, 8), (72, 4, 16), (64, 8, 16), (37, 4, 12), (20, 4, 4)])
def test_plan_right_looking_panels(n, tile, panel):
    """plan.h: potrf_rl / forward_solve_rl.  The update of the next block column is a full rectangle: its tiles above
    the diagonal read the (never used) upper part of A -- finite junk here, as on the device -- and are dead writes."""
    rng = np.random.default_rng(n + tile + panel)
    A0 = _spd(n, rng)
    A = bm.pad_identity(A0, tile)
    npad = A.shape[0]
    Aref = A.copy()
    A[np.triu_indices(npad, 1)] = 12345.0  # junk, not NaN: the rectangle update reads and rewrites upper tiles
    W = np.full((npad, npad), np.nan)
    for o in range(0, npad, tile):
        W[o:o + tile, o:o + tile] = 0
    T = np.full((npad, npad), np.nan)
    log = {}
    assert bm.potrf_rl(A, W, T, npad, tile, panel, log) == 0
    Lref = np.linalg.cholesky(Aref)
    assert np.allclose(np.diag(A), np.diag(Lref), rtol=1e-12)
    r = rng.standard_normal(npad)
    z = r.copy()
    bm.forward_solve_rl(T, W, z, npad, panel)
    assert np.allclose(z, np.linalg.solve(Lref, r), rtol=1e-9, atol=1e-12)
    if npad // panel >= 4:
        assert log["flops"] < 1.6 * npad**3 / 3

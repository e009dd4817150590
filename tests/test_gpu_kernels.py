"""GPU parity tests of the individual HIP kernels, called through the C ABI
(include/gpcore.h) via ctypes.  Reference for every check: NumPy fp64."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gpyreg_amd import _lib

    return _lib.context(0)


def _ref_gemm(A, B, C, M, N, K, akm, bkm, alpha, beta, klo, khi, lower, tile=128):
    out = C.copy()
    for ti in range(M // tile):
        for tj in range(N // tile):
            if lower and tj > ti:
                continue
            k0 = {0: 0, 1: ti * tile, 2: tj * tile}[klo]
            k1 = min({0: K, 1: (ti + 1) * tile, 2: (tj + 1) * tile}[khi], K)
            r, c = slice(ti * tile, (ti + 1) * tile), slice(tj * tile, (tj + 1) * tile)
            acc = np.zeros((tile, tile))
            if k1 > k0:
                a = A[k0:k1, r].T if akm else A[r, k0:k1]
                b = B[k0:k1, c] if bkm else B[c, k0:k1].T
                acc = a @ b
            out[r, c] = (beta * C[r, c] if beta else 0) + alpha * acc
    return out


@pytest.mark.parametrize("bt", [128, 64])
@pytest.mark.parametrize("akm,bkm", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize(
    "M,N,K,klo,khi,lower,alpha,beta",
    [
        (128, 128, 128, 0, 0, 0, 1.0, 0),
        (256, 384, 128, 0, 0, 0, -1.0, 1),
        (384, 384, 384, 0, 0, 1, -1.0, 1),   # syrk-like
        (384, 256, 256, 0, 2, 0, 1.0, 0),    # k <= col tile (trsm as product)
        (256, 384, 384, 2, 0, 0, 1.0, 0),    # k >= col tile
        (384, 256, 384, 0, 1, 0, -1.0, 0),   # k <= row tile
        (384, 384, 384, 1, 0, 1, 1.0, 0),    # lauum: k >= row tile, lower tiles
    ],
)
def test_gemm_modes_fp64(ctx, bt, akm, bkm, M, N, K, klo, khi, lower, alpha, beta):
    rng = np.random.default_rng(M + 3 * N + 7 * K + 11 * klo + 13 * khi + akm * 17 + bkm * 19)
    A = rng.standard_normal((K, M) if akm else (M, K))
    B = rng.standard_normal((K, N) if bkm else (N, K))  # asymmetric operands on purpose
    C0 = rng.standard_normal((M, N))
    got = ctx.debug_gemm(A, B, C0, M, N, K, akm, bkm, alpha, beta, klo, khi, lower, force_bt=bt)
    ref = _ref_gemm(A, B, C0, M, N, K, akm, bkm, alpha, beta, klo, khi, lower)
    mask = np.ones((M, N), bool)
    if lower:  # tiles strictly above the diagonal (in units of the kernel's tile) are skipped
        for ti in range(M // bt):
            for tj in range(ti + 1, N // bt):
                mask[ti * bt:(ti + 1) * bt, tj * bt:(tj + 1) * bt] = False
    err = np.abs(got - ref)[mask].max()
    assert err < 1e-11 * max(1.0, np.abs(ref[mask]).max()), err
    # untouched tiles of a lower_only launch keep their input value
    if lower:
        assert np.array_equal(got[~mask], C0[~mask])


@pytest.mark.experiments
@pytest.mark.parametrize("akm,bkm", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize(
    "M,N,K,klo,khi,lower,alpha,beta",
    [
        (128, 128, 128, 0, 0, 0, 1.0, 0),
        (256, 384, 128, 0, 0, 0, -1.0, 1),
        (384, 384, 384, 0, 0, 1, -1.0, 1),
        (384, 256, 256, 0, 2, 0, 1.0, 0),
        (256, 384, 384, 2, 0, 0, 1.0, 0),
        (384, 256, 384, 0, 1, 0, -1.0, 0),
        (640, 640, 640, 1, 0, 1, 1.0, 0),
    ],
)
def test_rectangular_tile_gives_the_bits_of_the_square_tiles(ctx, akm, bkm, M, N, K, klo, khi, lower, alpha, beta):
    """The 128 x 64 tile (gemm.h: BTN; round 5) against the 64-tile launch of the same product: every element keeps its
    k-range and k-order, so the results are identical bit for bit -- fp64 and fp32, every operand orientation, every
    k-range mode, lower-only launches (where the rectangular launch covers the 64-tile launch's tiles and, above the
    diagonal of a 128-row block, one 64-tile more per row of tiles: compared on the 64-tile launch's tiles)."""
    from gpyreg_amd import _lib

    rng = np.random.default_rng(M + 3 * N + 7 * K + 11 * klo + 13 * khi + akm * 17 + bkm * 19 + 23)
    A = rng.standard_normal((K, M) if akm else (M, K))
    B = rng.standard_normal((K, N) if bkm else (N, K))
    C0 = rng.standard_normal((M, N))
    for dtype in (_lib.F64, _lib.F32):
        sq = ctx.debug_gemm(A, B, C0, M, N, K, akm, bkm, alpha, beta, klo, khi, lower, dtype=dtype, force_bt=64)
        rc = ctx.debug_gemm(A, B, C0, M, N, K, akm, bkm, alpha, beta, klo, khi, lower, dtype=dtype, force_bt=12864)
        mask = np.ones((M, N), bool)
        if lower:
            for ti in range(M // 64):
                for tj in range(ti + 1, N // 64):
                    mask[ti * 64:(ti + 1) * 64, tj * 64:(tj + 1) * 64] = False
        assert np.array_equal(sq[mask], rc[mask]), (dtype, np.abs(sq - rc)[mask].max())


def test_gemm_fp32(ctx):
    from gpyreg_amd import _lib

    rng = np.random.default_rng(5)
    M = N = K = 256
    A, B = rng.standard_normal((M, K)), rng.standard_normal((K, N))
    ref = A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64)
    for bt in (128, 64):
        got = ctx.debug_gemm(A, B, np.zeros((M, N)), M, N, K, 0, 1, dtype=_lib.F32, force_bt=bt)
        assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()
        got = ctx.debug_gemm(B.T.copy(), A.T.copy(), np.zeros((N, M)), N, M, K, 0, 1, dtype=_lib.F32, force_bt=bt)
        assert np.abs(got - ref.T).max() < 2e-4 * np.abs(ref).max()


def _spd(n, rng, noise=0.01, D=2):
    X = rng.uniform(-3, 3, (n, D))
    d = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d / D) / noise + np.eye(n)


@pytest.mark.parametrize("n", [128, 7, 16, 17, 33, 100, 112, 113, 129, 256, 300, 640])
def test_factor_inverse_fp64(ctx, n):
    rng = np.random.default_rng(n)
    A = _spd(n, rng)
    L, W, Ainv, logdet, info = ctx.debug_factor(A)
    assert info == 0
    Lref = np.linalg.cholesky(A)
    assert np.allclose(L, Lref, rtol=1e-10, atol=1e-12)
    assert np.allclose(W, np.linalg.inv(Lref), rtol=1e-8, atol=1e-11)
    assert np.all(np.triu(W, 1) == 0)
    ref_inv = np.linalg.inv(A)
    il = np.tril_indices(n)
    assert np.allclose(Ainv[il], ref_inv[il], rtol=1e-7, atol=1e-11)
    assert abs(logdet - np.log(np.diag(Lref)).sum()) < 1e-10 * max(1, abs(logdet))


@pytest.mark.parametrize("n", [128, 5, 16, 17, 100, 113, 300])
def test_pipelined_leaf_is_bit_identical_to_the_phase_ordered_leaf(ctx, n):
    """leaf5 (wave-specialised, software-pipelined) performs the same MFMA sequence on the same operands as
    leaf3 (one phase after the other): L, W = L^-1, log det and info must agree bit for bit, fp64 and fp32."""
    from gpyreg_amd import _lib

    rng = np.random.default_rng(1000 + n)
    A = _spd(n, rng)
    for dtype in (_lib.F64, _lib.F32):
        out = {}
        for leaf in (3, 5):
            ctx.set_option("leaf", leaf)
            out[leaf] = ctx.debug_factor(A, want_inv=False, dtype=dtype)
        ctx.set_option("leaf", 5)
        (L3, W3, _, ld3, i3), (L5, W5, _, ld5, i5) = out[3], out[5]
        assert i3 == 0 and i5 == 0
        assert np.array_equal(L3, L5) and np.array_equal(W3, W5) and ld3 == ld5
    A[n // 2, n // 2] = -1.0  # the same failed pivot is reported
    infos = []
    for leaf in (3, 5):
        ctx.set_option("leaf", leaf)
        infos.append(ctx.debug_factor(A, want_inv=False)[-1])
    ctx.set_option("leaf", 5)
    assert infos[0] == infos[1] == n // 2 + 1


def test_factor_reports_non_pd(ctx):
    A = np.eye(200)
    A[150, 150] = -1.0
    *_, info = ctx.debug_factor(A, want_inv=False)
    assert info == 151
    A = np.eye(64)
    A[3, 3] = np.nan
    *_, info = ctx.debug_factor(A, want_inv=False)
    assert info == 4


def test_factor_fp32(ctx):
    from gpyreg_amd import _lib

    rng = np.random.default_rng(9)
    A = _spd(300, rng, noise=0.5)
    L, W, Ainv, logdet, info = ctx.debug_factor(A, dtype=_lib.F32)
    assert info == 0
    Lref = np.linalg.cholesky(A)
    assert np.abs(L - Lref).max() < 1e-4 * np.abs(Lref).max()
    assert abs(logdet - np.log(np.diag(Lref)).sum()) < 1e-4 * abs(logdet)


def test_mfma_peak_reports(ctx):
    from gpyreg_amd import _lib

    t64, c64, g64 = ctx.mfma_peak(_lib.F64)
    t32, c32, g32 = ctx.mfma_peak(_lib.F32)
    print("MFMA ceiling fp64: %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at %.2f GHz" % (t64, c64, g64))
    print("MFMA ceiling fp32: %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at %.2f GHz" % (t32, c32, g32))
    v64, vc64, _ = ctx.mfma_peak(2)
    v32, vc32, _ = ctx.mfma_peak(3)
    print("VALU FMA ceiling fp64: %.1f TFLOP/s, %.1f cycles/wave-instr/SIMD; fp32: %.1f TFLOP/s, %.1f cycles" % (v64, vc64, v32, vc32))
    assert t64 > 10 and t32 > 20

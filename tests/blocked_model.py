"""NumPy model of the device algorithm (test helper, CPU only).

The HIP library factors A = L L^T and forms W = L^-1 and A^-1 = W^T W using ONE
tiled GEMM primitive with per-tile k-ranges plus a small in-LDS leaf.  This
model executes exactly that launch plan (same primitive signature, same modes,
same buffer roles A / W / T) with NumPy tiles, so that the plan itself --
recursion order, triangular k-ranges, in-place hazards, identity padding -- is
verified on the CPU.  `gpyreg_amd/csrc/gpcore_plan.h` implements the same plan.
"""

import numpy as np

KLO_ZERO, KLO_ROW, KLO_COL = 0, 1, 2
KHI_FULL, KHI_ROW, KHI_COL = 0, 1, 2


def tiled_gemm(C, A, B, M, N, K, tile, *, a_kmajor, b_kmajor, alpha, beta,
               klo=KLO_ZERO, khi=KHI_FULL, lower_only=False, log=None):
    """C[M,N] = beta*C + alpha * Aop @ Bop computed tile by tile.

    a_kmajor False: A is stored (M,K);  True: stored (K,M).
    b_kmajor False: B is stored (N,K);  True: stored (K,N).
    Each output tile (ti,tj) only sums k in [k0,k1) chosen by klo/khi.
    Inputs are snapshotted per launch like a GPU would read them (the caller
    must not alias C with A or B unless every tile only reads what it owns).
    """
    assert M % tile == 0 and N % tile == 0 and K % tile == 0
    Ain, Bin = A.copy(), B.copy()  # catches accidental in-place aliasing below
    aliased = np.shares_memory(C, A) or np.shares_memory(C, B)
    assert not aliased, "plan must never alias output with an input"
    for ti in range(M // tile):
        for tj in range(N // tile):
            if lower_only and tj > ti:
                continue
            k0 = {KLO_ZERO: 0, KLO_ROW: ti * tile, KLO_COL: tj * tile}[klo]
            k1 = {KHI_FULL: K, KHI_ROW: (ti + 1) * tile, KHI_COL: (tj + 1) * tile}[khi]
            k1 = min(k1, K)
            r = slice(ti * tile, (ti + 1) * tile)
            c = slice(tj * tile, (tj + 1) * tile)
            acc = np.zeros((tile, tile))
            if k1 > k0:
                a = Ain[k0:k1, r].T if a_kmajor else Ain[r, k0:k1]
                b = Bin[k0:k1, c] if b_kmajor else Bin[c, k0:k1].T
                acc = a @ b
            C[r, c] = (beta * C[r, c] if beta != 0 else 0) + alpha * acc
            if log is not None:
                log["flops"] = log.get("flops", 0) + 2 * tile * tile * max(k1 - k0, 0)


def leaf(Ablk, Wblk):
    """tile x tile Cholesky + inverse, in place (device: one workgroup in LDS).
    Writes L into the lower triangle of Ablk (upper left untouched = garbage) and
    the full tile of Wblk (zeros above the diagonal).  Returns info (0 = ok)."""
    n = Ablk.shape[0]
    L = np.tril(Ablk).copy()
    for j in range(n):
        d = L[j, j]
        if not (d > 0):
            return j + 1
        d = np.sqrt(d)
        L[j, j] = d
        L[j + 1:, j] /= d
        for k in range(j + 1, n):
            L[k:, k] -= L[k:, j] * L[k, j]
    Ablk[np.tril_indices(n)] = L[np.tril_indices(n)]
    Wblk[:, :] = np.linalg.solve(L, np.eye(n))
    Wblk[np.triu_indices(n, 1)] = 0.0
    return 0


def leaf_panels(Ablk, Wblk, pw=16, refine=False):
    """The device leaf's own arithmetic (gpyreg_amd/csrc/leaf.h): right-looking with `pw`-wide panels, the panel
    solve L_iP = A_iP W_PP^T as a PRODUCT with the explicit inverse of the pw x pw diagonal block; with `refine`
    (the device's stable mode) followed by one step of refinement, L_iP += (A_iP - L_iP L_PP^T) W_PP^T."""
    n = Ablk.shape[0]
    S = np.tril(Ablk) + np.tril(Ablk, -1).T
    L = np.zeros((n, n))
    for p in range(0, n, pw):
        P = slice(p, min(p + pw, n))
        m = P.stop - P.start
        Lpp = np.tril(S[P, P]).copy()
        for j in range(m):  # pivot by reciprocal square root, column scaled by multiplication (leaf.h: diag16)
            d = Lpp[j, j]
            if not (d > 0):
                return p + j + 1
            r = 1.0 / np.sqrt(d)
            Lpp[j:, j] *= r
            for k in range(j + 1, m):
                Lpp[k:, k] -= Lpp[k:, j] * Lpp[k, j]
        L[P, P] = Lpp
        if P.stop < n:
            R = slice(P.stop, n)
            Wpp = np.linalg.solve(Lpp, np.eye(m))
            Lr = S[R, P] @ Wpp.T
            if refine:
                Lr = Lr + (S[R, P] - Lr @ Lpp.T) @ Wpp.T
            L[R, P] = Lr
            S[R, R] -= Lr @ Lr.T
    Ablk[np.tril_indices(n)] = L[np.tril_indices(n)]
    if refine:
        Ablk[np.triu_indices(n, 1)] = 0.0  # stable mode: L11 becomes a GEMM operand, its diagonal tiles must be clean
    Wblk[:, :] = np.linalg.solve(L, np.eye(n))
    Wblk[np.triu_indices(n, 1)] = 0.0
    return 0


def potrf_inv(A, W, T, off, n, tile, need_inv, post_mode, log=None, stable=False, leaf_fn=None):
    """Recursive plan on the n x n diagonal block starting at `off`.

    ``stable`` (the device's mode for jitter retries, plan.h): the trsm-as-a-product T21 = A21 W11^T is followed by
    one step of refinement against the factor itself, T21 += (A21 - T21 L11^T) W11^T, which makes the panel as
    accurate as a triangular solve when L11 is ill-conditioned (tests/analysis/jitter_model.py); L21 is kept in A at every
    node so that L11 is a complete operand."""
    post_mode = post_mode or stable
    if n == tile:
        s = slice(off, off + n)
        info = (leaf_fn or leaf)(A[s, s], W[s, s])
        if log is not None:
            log["launches"] = log.get("launches", 0) + 1
        return info + off if info else 0
    q = n // tile
    n1 = (q // 2) * tile if q > 1 else tile
    n2 = n - n1
    o1, o2 = off, off + n1
    r1, r2 = slice(o1, o1 + n1), slice(o2, o2 + n2)
    info = potrf_inv(A, W, T, o1, n1, tile, True, post_mode, log, stable, leaf_fn)
    if info:
        return info
    # step 2: T21 = A21 * W11^T      (W11 lower: k <= col)
    tiled_gemm(T[r2, r1], A[r2, r1], W[r1, r1], n2, n1, n1, tile, a_kmajor=False,
               b_kmajor=False, alpha=1.0, beta=0.0, khi=KHI_COL, log=log)
    if stable:
        # step 2b: R = A21 - T21 * L11^T (in place in A21; L11 lower: k <= col), T21 += R * W11^T
        tiled_gemm(A[r2, r1], T[r2, r1], A[r1, r1], n2, n1, n1, tile, a_kmajor=False,
                   b_kmajor=False, alpha=-1.0, beta=1.0, khi=KHI_COL, log=log)
        tiled_gemm(T[r2, r1], A[r2, r1], W[r1, r1], n2, n1, n1, tile, a_kmajor=False,
                   b_kmajor=False, alpha=1.0, beta=1.0, khi=KHI_COL, log=log)
    # step 3: A22 -= T21 * T21^T     (lower tiles only)
    tiled_gemm(A[r2, r2], T[r2, r1], T[r2, r1], n2, n2, n1, tile, a_kmajor=False,
               b_kmajor=False, alpha=-1.0, beta=1.0, lower_only=True, log=log)
    info = potrf_inv(A, W, T, o2, n2, tile, need_inv, post_mode, log, stable, leaf_fn)
    if info:
        return info
    if need_inv:
        # step 5a: U = T21 * W11      (W11 lower: k >= col); U lives in A21
        tiled_gemm(A[r2, r1], T[r2, r1], W[r1, r1], n2, n1, n1, tile, a_kmajor=False,
                   b_kmajor=True, alpha=1.0, beta=0.0, klo=KLO_COL, log=log)
        # step 5b: W21 = -W22 * U     (W22 lower: k <= row)
        tiled_gemm(W[r2, r1], W[r2, r2], A[r2, r1], n2, n1, n2, tile, a_kmajor=False,
                   b_kmajor=True, alpha=-1.0, beta=0.0, khi=KHI_ROW, log=log)
    if post_mode:
        A[r2, r1] = T[r2, r1]  # keep L21 in A (posterior fetch); forward_solve reads it from the scratch
    if log is not None:
        log["launches"] = log.get("launches", 0) + (4 if need_inv else 2)
    return 0


def lauum(Kinv, W, n, tile, log=None):
    """Kinv = W^T W, lower tiles, k >= row tile."""
    tiled_gemm(Kinv, W, W, n, n, n, tile, a_kmajor=True, b_kmajor=True, alpha=1.0,
               beta=0.0, klo=KLO_ROW, lower_only=True, log=log)


def pad_identity(Amat, tile):
    n = Amat.shape[0]
    npad = -(-n // tile) * tile
    P = np.eye(npad)
    P[:n, :n] = Amat
    return P


def forward_solve(T, W, r, off, n, tile, need_inv):
    """z = L^-1 r using what potrf_inv(need_inv=...) left behind: a block that has
    its full inverse multiplies by W (trmv); otherwise split like the factorization
    and use L21, still in the scratch T where step 2 computed it."""
    if need_inv or n == tile:
        s = slice(off, off + n)
        r[s] = np.tril(W[s, s]) @ r[s]
        return
    q = n // tile
    n1 = (q // 2) * tile if q > 1 else tile
    n2 = n - n1
    r1, r2 = slice(off, off + n1), slice(off + n1, off + n)
    forward_solve(T, W, r, off, n1, tile, True)
    r[r2] -= T[r2, r1] @ r[r1]
    forward_solve(T, W, r, off + n1, n2, tile, need_inv)


def trsm_nll(A, W, T, r0, m, c0, n, tile, blk, log=None):
    """plan.h: trsm_nll.  T[r0:, c0:] = A[r0:, c0:] L[c0:, c0:]^-T by blocks: blocks of at most `blk` rows own an
    inverse, above that the solve splits like the factorization (A21 is updated in place)."""
    rows = slice(r0, r0 + m)
    if n <= blk:
        c = slice(c0, c0 + n)
        tiled_gemm(T[rows, c], A[rows, c], W[c, c], m, n, n, tile, a_kmajor=False, b_kmajor=False, alpha=1.0,
                   beta=0.0, khi=KHI_COL, log=log)
        return
    q = n // tile
    n1 = (q // 2) * tile
    n2 = n - n1
    ca, cc = slice(c0, c0 + n1), slice(c0 + n1, c0 + n)
    trsm_nll(A, W, T, r0, m, c0, n1, tile, blk, log)
    tiled_gemm(A[rows, cc], T[rows, ca], T[cc, ca], m, n2, n1, tile, a_kmajor=False, b_kmajor=False, alpha=-1.0,
               beta=1.0, log=log)
    trsm_nll(A, W, T, r0, m, c0 + n1, n2, tile, blk, log)


def potrf_nll(A, W, T, off, n, tile, blk, log=None):
    """plan.h: potrf_nll -- the factorization of an NLL-only evaluation at N^3/3."""
    if n <= blk or n == tile:
        return potrf_inv(A, W, T, off, n, tile, True, False, log)
    q = n // tile
    n1 = (q // 2) * tile
    n2 = n - n1
    o1, o2 = off, off + n1
    r1, r2 = slice(o1, o1 + n1), slice(o2, o2 + n2)
    info = potrf_nll(A, W, T, o1, n1, tile, blk, log)
    if info:
        return info
    trsm_nll(A, W, T, o2, n2, o1, n1, tile, blk, log)
    tiled_gemm(A[r2, r2], T[r2, r1], T[r2, r1], n2, n2, n1, tile, a_kmajor=False, b_kmajor=False, alpha=-1.0,
               beta=1.0, lower_only=True, log=log)
    return potrf_nll(A, W, T, o2, n2, tile, blk, log)


def forward_solve_nll(T, W, r, off, n, tile, blk):
    if n <= blk or n == tile:
        s = slice(off, off + n)
        r[s] = np.tril(W[s, s]) @ r[s]
        return
    q = n // tile
    n1 = (q // 2) * tile
    n2 = n - n1
    r1, r2 = slice(off, off + n1), slice(off + n1, off + n)
    forward_solve_nll(T, W, r, off, n1, tile, blk)
    r[r2] -= T[r2, r1] @ r[r1]
    forward_solve_nll(T, W, r, off + n1, n2, tile, blk)


def potrf_rl(A, W, T, npad, tile, panel, log=None):
    """plan.h: potrf_rl -- right-looking panels (the look-ahead only reorders launches).  Per panel: D (potrf_inv of
    the diagonal block), P (panel solve into the scratch), N (update of the next block column, a full rectangle whose
    tiles above the diagonal are dead writes), R (lower tiles of the rest)."""
    for o in range(0, npad, panel):
        nb = min(panel, npad - o)
        rest = npad - o - nb
        info = potrf_inv(A, W, T, o, nb, tile, True, False, log)
        if info:
            return info
        if rest == 0:
            break
        d, below = slice(o, o + nb), slice(o + nb, npad)
        tiled_gemm(T[below, d], A[below, d], W[d, d], rest, nb, nb, tile, a_kmajor=False, b_kmajor=False, alpha=1.0,
                   beta=0.0, khi=KHI_COL, log=log)
        nb2 = min(panel, rest)
        rest2 = rest - nb2
        nxt = slice(o + nb, o + nb + nb2)
        tiled_gemm(A[below, nxt], T[below, d], T[nxt, d], rest, nb2, nb, tile, a_kmajor=False, b_kmajor=False,
                   alpha=-1.0, beta=1.0, log=log)
        if rest2 > 0:
            r2 = slice(o + nb + nb2, npad)
            tiled_gemm(A[r2, r2], T[r2, d], T[r2, d], rest2, rest2, nb, tile, a_kmajor=False, b_kmajor=False,
                       alpha=-1.0, beta=1.0, lower_only=True, log=log)
    return 0


def forward_solve_rl(T, W, r, npad, panel):
    for o in range(0, npad, panel):
        nb = min(panel, npad - o)
        d, below = slice(o, o + nb), slice(o + nb, npad)
        r[d] = np.tril(W[d, d]) @ r[d]
        if o + nb < npad:
            r[below] -= T[below, d] @ r[d]

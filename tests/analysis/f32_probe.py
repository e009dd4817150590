"""Where does the fp32 mode lose accuracy?  Device factor / inverse of one N=1408 RQ system in
fp32 through the debug hooks, decomposed against fp64 LAPACK on the host (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as sla
from oracle import gp_oracle as orc
from gpyreg_amd import _lib

def setup(kernel, degree, N, D):
    rng = np.random.default_rng(N + degree + len(kernel))
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(X.sum(1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
    model = dict(kernel=kernel, degree=degree, mean="const", noise=(1, 0, 0))
    cov_N = orc.cov_count(kernel, D); nl = D if cov_N > 2 else 1
    base = np.concatenate([np.log(1.5 * np.sqrt(D)) * np.ones(nl), np.zeros(cov_N - nl), [np.log(0.1)], [0.0]])
    hyp = base + 0.1 * rng.standard_normal((10, base.size))
    return model, X, y, hyp[9], cov_N

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1408
model, X, y, h, cov_N = setup("rq", 0, N, 3)
rn, rd = orc.core(model, h, X, y, None, 1, 1)
K, dK = orc.covariance("rq", h[:cov_N], X, compute_grad=True)
sn2 = np.exp(2 * h[cov_N]); r = y - h[-1]
A = K / sn2 + np.eye(N)
sc = np.abs(rd[:cov_N]).max()
ctx = _lib.context(0)
Lx = sla.cholesky(A, lower=True); Wx = sla.solve_triangular(Lx, np.eye(N), lower=True); Ainv_x = Wx.T @ Wx

def grad_from(Ainv, alpha):
    Q = Ainv / sn2 - alpha @ alpha.T
    return np.array([np.sum(Q * dK[:, :, i]) / 2 for i in range(cov_N)])

for dt, nm in ((_lib.F64, "f64"), (_lib.F32, "f32")):
    L, W, Ainv, logdet, info = ctx.debug_factor(A, want_inv=True, dtype=dt)
    Ainv = np.tril(Ainv) + np.tril(Ainv, -1).T
    print(nm, "info", info, "|L-Lx|/|Lx| %.2e" % (np.abs(L - Lx).max() / np.abs(Lx).max()),
          "|W-Wx|/|Wx| %.2e" % (np.abs(W - Wx).max() / np.abs(Wx).max()),
          "|WL-I| %.2e" % np.abs(W @ Lx - np.eye(N)).max(),
          "|Ainv - WtW(f64 of dev W)|/|Ainv| %.2e" % (np.abs(Ainv - W.T @ W).max() / np.abs(Ainv_x).max()),
          "|Ainv-Ainv_x|/|.| %.2e" % (np.abs(Ainv - Ainv_x).max() / np.abs(Ainv_x).max()))
    a1 = (W.T @ (W @ r)) / sn2
    print("   grad err: dev W, host f64 products %.2e | dev Ainv + that alpha %.2e | exact alpha + dev Ainv %.2e" % (
        np.abs(grad_from(W.T @ W, a1) - rd[:cov_N]).max() / sc, np.abs(grad_from(Ainv, a1) - rd[:cov_N]).max() / sc,
        np.abs(grad_from(Ainv, Ainv_x @ r / sn2) - rd[:cov_N]).max() / sc))
# LAPACK fp32 for comparison
A32 = A.astype(np.float32); L32 = sla.cholesky(A32, lower=True); W32 = sla.solve_triangular(L32, np.eye(N, dtype=np.float32), lower=True)
Ai32 = (W32.T @ W32).astype(float)
print("lapack f32: |W-Wx| %.2e |Ainv-Ainv_x| %.2e  grad err (all f32 products) %.2e" % (
    np.abs(W32 - Wx).max() / np.abs(Wx).max(), np.abs(Ai32 - Ainv_x).max() / np.abs(Ainv_x).max(),
    np.abs(grad_from(Ai32, (W32.T.astype(float) @ (W32.astype(float) @ r)) / sn2) - rd[:cov_N]).max() / sc))

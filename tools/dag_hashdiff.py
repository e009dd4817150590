"""Which 128-tiles of the workspace differ between the stream-ordered schedule and the dataflow graph?
usage: python tools/dag_hashdiff.py N S grad [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

N, S, grad = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
dtype = sys.argv[4] if len(sys.argv) > 4 else "f64"
ctx = _lib.context(0)
bench.CONFIGS[3] = dict(bench.CONFIGS[3], N=N)
X, y, hyp = bench.synthetic_problem(3, S)
gp = bench.make_gp(3, dtype)
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
nt = (N + 127) // 128


def hashes():
    out = {}
    for smp in range(S):
        for w, name in enumerate("AWT"):
            h = np.zeros(nt * nt, dtype=np.uint64)
            rc = ctx._lib.gpc_debug_workspace_hash(ctx._h, 0 if dtype == "f64" else 1, w, smp, h.ctypes.data)
            assert rc == 0
            out[name, smp] = h.reshape(nt, nt)
    return out


res = {}
for dag in (0, 1):
    ctx.set_option("dag", dag)
    for k in range(2):
        r = gp.nll_batch(hyp, compute_grad=grad)
    res[dag] = (r, hashes())
ctx.set_option("dag", 0)
print("nlZ equal per sample:", (res[0][0][0] == res[1][0][0]).tolist())
low = np.tril(np.ones((nt, nt), bool), -1)
for smp in range(S):
    for name in "AWT":
        d = res[0][1][name, smp] != res[1][1][name, smp]
        print("sample", smp, name, "tiles that differ strictly below the diagonal:",
              [tuple(int(v) for v in ij) for ij in np.argwhere(d & low)][:30], " on the diagonal:", int(np.diag(d).sum()),
              " above:", int((d & ~low).sum() - np.diag(d).sum()))

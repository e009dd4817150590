"""Do independent single-sample pipelines overlap usefully on one GPU?  K host threads, each with its OWN library
context (stream + workspace) on the same device, each evaluating ONE sample per call (launch-graph replay), against one
context evaluating K samples per call in lock-step."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib
from gpyreg_amd.gaussian_process import _DTYPES

cfg = 3
grad = "--nll" not in sys.argv
for K in (2, 4, 8):
    X, y, hyp = bench.synthetic_problem(cfg, K)
    gp = bench.make_gp(cfg, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    kid, deg = gp._kid()
    cov_N = gp._counts()[0]
    for _ in range(3):
        gp.nll_batch(hyp, grad)
    t0 = time.perf_counter()
    for _ in range(10):
        gp.nll_batch(hyp, grad)
    t_lock = (time.perf_counter() - t0) / 10
    ctxs = [_lib.Context(0) for _ in range(K)]
    for c in ctxs:
        c.set_data(X, y)
    pvs = [gp._plugin_values(hyp[k:k + 1], grad) for k in range(K)]
    outs = [None] * K
    def work(k, reps):
        pv = pvs[k]
        for _ in range(reps):
            outs[k] = ctxs[k].nll_batch(kid, deg, _DTYPES[gp.dtype], hyp[k:k + 1, :cov_N], pv["m"], pv["sn2"], pv["vec"], grad, pv["dm"], pv["dsn2"])
    for k in range(K):
        work(k, 2)
    th = [threading.Thread(target=work, args=(k, 10)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t_par = (time.perf_counter() - t0) / 10
    ref = gp.nll_batch(hyp, grad)[0]
    same = all(outs[k][0][0] == ref[k] for k in range(K))
    print(f"N=4096 {'NLL+grad' if grad else 'NLL'} K={K}: lock-step batch {t_lock*1e3:.3f} ms   {K} independent contexts {t_par*1e3:.3f} ms   same bits: {same}", flush=True)
    for c in ctxs:
        c.close()

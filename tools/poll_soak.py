"""Soak of the completion-by-polled-word path (round 6): tens of thousands of calls below N_pad = 2048, sizes and modes
interleaved (one-leaf NLL, one-leaf gradient, general pipeline NLL / gradient, posterior + predict in between), from
one thread and from two threads with a GP each -- every result must equal the first result of its kind BIT FOR BIT, the
polled / synchronised counters are printed, and so is the process's resident set before and after (the stream is never
synchronised by the polled calls: nothing may pile up in the runtime).   usage: python tools/poll_soak.py [rounds=4000]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib


def rss_mb():
    with open("/proc/self/status") as fh:
        for line in fh:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    return float("nan")


def make(N, S, seed):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    return gp, hyp


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ctx = _lib.context()
cases = [make(50, 1, 0), make(128, 3, 1), make(200, 1, 2), make(700, 2, 3), make(1500, 1, 4)]
ref = [(gp.nll_batch(h, False)[0].copy(), tuple(a.copy() for a in gp.nll_batch(h, True))) for gp, h in cases]
r0 = rss_mb()
t0 = time.perf_counter()
bad = 0
for it in range(rounds):
    for k, (gp, h) in enumerate(cases):
        n0 = gp.nll_batch(h, False)[0]
        n1, d1 = gp.nll_batch(h, True)
        if not (np.array_equal(n0, ref[k][0]) and np.array_equal(n1, ref[k][1][0]) and np.array_equal(d1, ref[k][1][1])):
            bad += 1
    if it % 500 == 499:
        gp, h = cases[3]
        gp.update(hyp=h)  # a posterior build and a prediction in between (these wait for the stream)
        gp.predict(gp.X[:5])
        print(f"round {it + 1}: mismatches {bad}, polled {ctx.get_option('small_polled')}, synchronised {ctx.get_option('small_synced')}, "
              f"RSS {rss_mb():.0f} MB (start {r0:.0f})", flush=True)
print(f"one thread: {rounds * len(cases) * 2} calls in {time.perf_counter() - t0:.1f} s, mismatches {bad}")

# two threads, a GP each (different data on the one context: the context lock serialises upload + call)
errs = []
def worker(k):
    gp, h = cases[k]
    for _ in range(rounds // 4):
        n0 = gp.nll_batch(h, False)[0]
        n1, d1 = gp.nll_batch(h, True)
        if not (np.array_equal(n0, ref[k][0]) and np.array_equal(n1, ref[k][1][0]) and np.array_equal(d1, ref[k][1][1])):
            errs.append(k)
ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 2)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"two threads: mismatches {len(errs)}; polled {ctx.get_option('small_polled')}, synchronised {ctx.get_option('small_synced')}, RSS {rss_mb():.0f} MB")
assert bad == 0 and not errs
print("poll soak OK")

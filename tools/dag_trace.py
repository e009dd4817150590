"""Where does the time of a dataflow graph go?  Runs one evaluation with GPC_DAG_TRACE (per-task wall-clock stamps of sample 0:
ready, popped, started, ended, completed), joins them with the graph (gpc_debug_dag) and walks the chain of LAST-ARRIVING
predecessors back from the last task: the critical path as it actually ran, with its time split into
    queue    ready -> popped by a workgroup (ring latency, or waiting for a free workgroup)
    acquire  popped -> started (acquire fence, barriers, descriptor loads)
    run      the tile / the leaf, stores drained
    signal   ended -> the successor is ready (successor counters, push)
usage: python tools/dag_trace.py N S [key=value ...]   (GPU box)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

N, S = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:])
path = os.path.join(tempfile.gettempdir(), "dag_trace.txt")
os.environ["GPC_DAG_TRACE"] = path
import bench
from gpyreg_amd import _lib
import dag_model

ctx = _lib.context(0)
bench.CONFIGS[3] = dict(bench.CONFIGS[3], N=N)
X, y, hyp = bench.synthetic_problem(3, S)
gp = bench.make_gp(3, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
for k, v in opts.items():
    ctx.set_option(k, int(v))
ctx.set_option("dag", 1)
for _ in range(3):
    gp.nll_batch(hyp, compute_grad=True)
tr_all = np.loadtxt(path, dtype=np.int64)
npad = (N + 127) // 128 * 128
dag = dag_model.export(npad, 1 if int(opts.get("dag_lauum", 1)) else 2, 0, int(opts.get("dag_small_tiles", 40)))
tasks, succ = dag["tasks"], dag["succ"]
nt = tasks.shape[0]
assert tr_all.shape[0] == nt * S, (tr_all.shape, nt, S)
tr = tr_all[:nt]  # sample 0: the critical-path analysis
ready, start, pop, end, done, who = (tr[:, i] for i in range(1, 7))
kept = ready < 0
ready = np.abs(ready)
t0 = ready[ready > 0].min()
us = lambda v: (v - t0) / 100.0
preds = [[] for _ in range(nt)]
for t in range(nt):
    b, n = int(tasks[t, 21]), int(tasks[t, 22])
    for t2 in succ[b:b + n]:
        preds[int(t2)].append(t)
kind = np.where(tasks[:, 0] == 1, "leaf", np.where(tasks[:, 1] == 64, "t64", "t128"))
print(f"N={N} S={S}: {nt} tasks of sample 0; span {us(done.max()):.0f} us; kept (no ring) {int(kept.sum())}")
for k in ("leaf", "t64", "t128"):
    m = kind == k
    if not m.any():
        continue
    q = (pop - ready)[m] / 100.0
    a = (start - pop)[m] / 100.0
    r = (end - start)[m] / 100.0
    c = (done - end)[m] / 100.0
    print(f"  {k:5s} n={int(m.sum()):5d}  queue mean {q.mean():7.1f} med {np.median(q):6.1f} | acquire {a.mean():5.1f} | run mean {r.mean():7.1f} med {np.median(r):6.1f} | complete {c.mean():5.1f}")
# the critical path as it ran
t = int(np.argmax(done))
path_tasks = []
while True:
    path_tasks.append(t)
    if not preds[t]:
        break
    t = max(preds[t], key=lambda p: done[p])
path_tasks.reverse()
tot = dict(queue=0.0, acquire=0.0, run=0.0, signal=0.0)
cnt = dict(leaf=0, t64=0, t128=0)
for i, t in enumerate(path_tasks):
    cnt[kind[t]] += 1
    tot["queue"] += (pop[t] - ready[t]) / 100.0
    tot["acquire"] += (start[t] - pop[t]) / 100.0
    tot["run"] += (end[t] - start[t]) / 100.0
    if i + 1 < len(path_tasks):
        tot["signal"] += (ready[path_tasks[i + 1]] - end[t]) / 100.0
print(f"critical path as run: {len(path_tasks)} tasks ({cnt}), us by part: " + ", ".join(f"{k} {v:.0f}" for k, v in tot.items()),
      f"  sum {sum(tot.values()):.0f}")
per = {k: [] for k in ("leaf", "t64", "t128")}
for i, t in enumerate(path_tasks[:-1]):
    per[kind[t]].append(((pop[t] - ready[t]) / 100.0, (start[t] - pop[t]) / 100.0, (end[t] - start[t]) / 100.0,
                         (ready[path_tasks[i + 1]] - end[t]) / 100.0))
for k, v in per.items():
    if v:
        a = np.array(v)
        print(f"  on the path, {k:5s} n={len(v):4d}: queue {a[:,0].mean():6.1f}  acquire {a[:,1].mean():5.1f}  run {a[:,2].mean():7.1f}  signal {a[:,3].mean():6.1f}  (us, means)")
print("first hops of the path: task kind ready pop start end next-ready (us)")
for i, t in enumerate(path_tasks[:int(os.environ.get('DAG_TRACE_HOPS', 12))]):
    nxt = us(ready[path_tasks[i + 1]]) if i + 1 < len(path_tasks) else float("nan")
    print(f"  {t:6d} {kind[t]:5s} {'kept' if kept[t] else 'ring'} {us(ready[t]):9.1f} {us(pop[t]):9.1f} {us(start[t]):9.1f} {us(end[t]):9.1f} {nxt:9.1f}  w={int(who[t]) & 0xffffffff}/x{int(who[t]) >> 32}")

# occupancy of the launch over time, ALL samples: workgroups inside a task (start .. end), by kind, per time bin -- the timeline of
# how many of the 496 GEMM workgroups (and of the leaf servers) compute at any moment
kind_all = np.tile(kind, S)
st_all, en_all = tr_all[:, 2], tr_all[:, 4]
ok = (st_all > 0) & (en_all > 0)
T0, T1 = st_all[ok].min(), en_all[ok].max()
nb = 40
edges = np.linspace(T0, T1, nb + 1)
print(f"occupancy timeline, all {S} sample(s): {nb} bins of {(T1 - T0) / nb / 100.0:.0f} us; mean workgroups computing (t64 / t128 / leaf), of 496 + leaf servers")
for b in range(nb):
    lo, hi = edges[b], edges[b + 1]
    row = []
    for k in ("t64", "t128", "leaf"):
        m = ok & (kind_all == k)
        ov = np.clip(np.minimum(en_all[m], hi) - np.maximum(st_all[m], lo), 0, None).sum() / (hi - lo)
        row.append(ov)
    bar = "#" * int(round((row[0] + row[1]) / 496 * 60))
    print(f"  {(lo - T0) / 100.0:8.0f} us  {row[0]:6.1f} {row[1]:6.1f} {row[2]:4.1f}  |{bar}")

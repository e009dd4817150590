"""NumPy executor of the tile-task graph that the dataflow schedule runs on the device (test helper, CPU only).

``gpc_debug_dag`` (include/gpcore.h; host-only, no GPU needed) exports the graph gpyreg_amd/csrc/dag.h derives from the
recorded launches of plan.h: per task its operand regions in the three buffers A / W / T, its scalars, its number of
predecessors and its successor list.  ``run`` executes it exactly as the device does -- a task may start once its
predecessor count has reached zero, a finished task decrements its successors -- but picks the next task AT RANDOM
among the ready ones.  If an edge were missing (a read-after-write, write-after-read or write-after-write hazard the
host analysis overlooked), some order would read a tile too early or overwrite one too soon, and the buffers would
differ from those of the launch order: the test compares them bit for bit over several seeds.
"""

import ctypes as C

import numpy as np

from blocked_model import leaf_panels


def export(npad, plan, nll_blk=0, small_tiles=40):
    from gpyreg_amd import _lib

    lib = _lib.load()
    counts = np.zeros(4, dtype=np.int32)
    rc = lib.gpc_debug_dag(npad, plan, nll_blk, small_tiles, counts.ctypes.data, None, None, None, 0, 0)
    assert rc == 0, rc
    nt, ne = int(counts[0]), int(counts[1])
    tasks = np.zeros((nt, 24), dtype=np.int32)
    alpha = np.zeros(nt)
    succ = np.zeros(max(ne, 1), dtype=np.int32)
    rc = lib.gpc_debug_dag(npad, plan, nll_blk, small_tiles, counts.ctypes.data, tasks.ctypes.data, alpha.ctypes.data,
                           succ.ctypes.data, nt, max(ne, 1))
    assert rc == 0, rc
    return dict(tasks=tasks, alpha=alpha, succ=succ[:ne], counts=counts.copy(), npad=npad)


def _exec(task, alpha, bufs):
    is_leaf, bt, akm, bkm, beta = (int(v) for v in task[:5])
    cb, cr0, cr1, cc0, cc1 = (int(v) for v in task[5:10])
    if is_leaf:
        # the device leaf: L into the lower triangle of A's tile, W = L^-1 (whole tile, zeros above the diagonal)
        blk = bufs[0][cr0:cr1, cc0:cc1]
        a = np.tril(blk).copy()
        w = np.zeros_like(a)
        info = leaf_panels(a, w)
        assert info == 0
        il = np.tril_indices(a.shape[0])
        blk[il] = a[il]
        bufs[1][cr0:cr1, cc0:cc1] = w
        return
    ab, ar0, ar1, ac0, ac1 = (int(v) for v in task[10:15])
    bb, br0, br1, bc0, bc1 = (int(v) for v in task[15:20])
    acc = np.zeros((cr1 - cr0, cc1 - cc0))
    if ar1 > ar0 and ac1 > ac0:
        a = bufs[ab][ar0:ar1, ac0:ac1]
        b = bufs[bb][br0:br1, bc0:bc1]
        a = a.T if akm else a      # stored [k][m] -> (m, k)
        b = b if bkm else b.T      # stored [k][n] | [n][k] -> (k, n)
        acc = a @ b
    c = bufs[cb][cr0:cr1, cc0:cc1]
    c[...] = (c if beta else 0.0) + alpha * acc


def run(dag, A0, rng=None):
    """Execute the graph on a copy of the SPD matrix ``A0`` (npad x npad).  ``rng`` None: task-id order (= the launch
    order of plan.h); else a random valid order.  Returns (A, W, T) and the order taken.  Never-written cells hold
    NaN, so that a read of something no task produced shows."""
    npad = dag["npad"]
    tasks, succ = dag["tasks"], dag["succ"]
    nt = tasks.shape[0]
    bufs = [np.array(A0, dtype=float), np.full((npad, npad), np.nan), np.full((npad, npad), np.nan)]
    bufs[0][np.triu_indices(npad, 1)] = np.nan  # the strictly upper part of A is never read
    pending = tasks[:, 20].astype(np.int64).copy()
    ready = [t for t in range(nt) if pending[t] == 0]
    order = []
    while ready:
        if rng is None:
            k = int(np.argmin(ready))
        else:
            k = int(rng.integers(len(ready)))
        t = ready.pop(k)
        _exec(tasks[t], dag["alpha"][t], bufs)
        order.append(t)
        b, n = int(tasks[t, 21]), int(tasks[t, 22])
        for t2 in succ[b:b + n]:
            pending[t2] -= 1
            assert pending[t2] >= 0
            if pending[t2] == 0:
                ready.append(int(t2))
    assert len(order) == nt, "the graph has a cycle or an unreachable task"
    return bufs, order

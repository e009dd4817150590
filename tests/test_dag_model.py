"""The tile-task graph of the dataflow schedule (gpyreg_amd/csrc/dag.h), verified on the CPU: exported by the
host-only entry ``gpc_debug_dag`` and executed with NumPy tiles (tests/dag_model.py).

1. executed in launch order it IS the blocked factorization: diag L from the leaves, W = L^-1, T = A^-1 (lower tiles);
2. executed in random valid orders -- any order the device's ready rings could produce -- it gives the same bits in all
   three buffers: the dependency analysis (RAW / WAR / WAW on 64 x 64 cells) has no missing edge;
3. every plan variant the device uses: factor + inverse + W^T W, factor + inverse, NLL-only (left children inverted;
   blocked solves above 256 / 512 rows), with 64-tile, 128-tile and mixed cuts.
"""

import numpy as np
import pytest

import dag_model

pytestmark = pytest.mark.experiments  # gpc_debug_dag is exported by the experiments build only (conftest.py)


def _spd(n, nvalid, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, (nvalid, 3))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    A = np.eye(n)
    A[:nvalid, :nvalid] = np.exp(-0.5 * d2 / 1.5 ** 2) + 0.05 * np.eye(nvalid)
    return A


def _same(x, y):
    return all(np.array_equal(a, b, equal_nan=True) for a, b in zip(x, y))


@pytest.mark.parametrize("npad,plan,nll_blk,small", [
    (512, 1, 0, 40),     # everything 64-tiles
    (768, 1, 0, 4),      # mixed: launches of >= 4 128-tiles as 128-tiles
    (1024, 1, 0, 0),     # everything 128-tiles
    (1024, 2, 0, 8),
    (640, 0, 0, 40),     # NLL only, left children inverted (uneven split: 5 tiles)
    (1024, 0, 256, 6),   # NLL only, blocked solves above 256 rows
    (1536, 0, 512, 10),
])
def test_random_orders_give_the_bits_of_the_launch_order(npad, plan, nll_blk, small):
    dag = dag_model.export(npad, plan, nll_blk, small)
    nt, ne, nl, nleaf = (int(v) for v in dag["counts"])
    assert nleaf == npad // 128 and nt > nleaf
    A0 = _spd(npad, npad - 37, seed=npad + plan)
    ref, order0 = dag_model.run(dag, A0, None)
    assert order0 == sorted(order0)  # task ids are a topological order: the launch order of plan.h
    n = npad
    # (L21 does not survive in A: the inverse products overwrite it, and the scratch T ends as A^-1)
    assert abs(np.log(np.diag(ref[0])).sum() - 0.5 * np.linalg.slogdet(A0)[1]) < 1e-9  # diag L from the leaves
    if plan != 0:
        W = np.tril(ref[1])
        assert np.abs(W @ A0 @ W.T - np.eye(n)).max() < 1e-9
    if plan == 1:
        Ainv = np.tril(ref[2])
        Ainv = Ainv + np.tril(Ainv, -1).T
        assert np.abs(Ainv @ A0 - np.eye(n)).max() < 1e-7
    for seed in range(4):
        got, order = dag_model.run(dag, A0, np.random.default_rng(100 * seed + 7))
        assert order != order0
        assert _same(got, ref), (npad, plan, seed)


def test_the_graph_exposes_the_overlap_the_launch_order_hides():
    """Shape of the graph at N_pad = 2048: the inverse products U = T21 W11 are not on the path to the next leaf (the
    first leaf of a right child becomes ready while U tasks of its parent are still pending in launch order)."""
    dag = dag_model.export(2048, 1, 0, 40)
    tasks, succ = dag["tasks"], dag["succ"]
    nt = tasks.shape[0]
    # longest path in TASKS (not time) from the first leaf to the last one, against the number of tasks
    depth = np.zeros(nt, dtype=np.int64)
    for t in range(nt):
        b, n = int(tasks[t, 21]), int(tasks[t, 22])
        for t2 in succ[b:b + n]:
            depth[t2] = max(depth[t2], depth[t] + 1)
    leaves = np.flatnonzero(tasks[:, 0] == 1)
    assert len(leaves) == 16 and np.all(np.diff(depth[leaves]) > 0)
    assert depth.max() < nt / 20  # thousands of tasks, a chain of a few hundred: the rest can overlap

"""world_size-2 CPU (gloo) test of the multi-GPU path: sample sharding + all-gather.

The device evaluator is replaced by a stand-in that calls the CPU oracle (tests may);
what is under test is gpyreg_amd.sharding: partition, padding of unequal shards, the
all-gather and the reassembly order."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleGP:
    """Quacks like gpyreg_amd.GP.nll_batch / predict, computed by the oracle."""

    def __init__(self, model, X, y):
        self.model, self.X, self.y = model, X, y
        self.calls = []
        self.posts = None

    def nll_batch(self, hyp, compute_grad=False):
        from oracle import gp_oracle as orc

        self.calls.append(hyp.shape[0])
        nl, dn = [], []
        for h in hyp:
            r = orc.core(self.model, h, self.X, self.y, None, 1, 1 if compute_grad else 0)
            nl.append(r[0] if compute_grad else r)
            if compute_grad:
                dn.append(r[1])
        return np.array(nl), (np.stack(dn) if compute_grad else None)

    def predict(self, xs, separate_samples=True):
        from oracle import gp_oracle as orc

        return orc.predict(self.model, self.posts, self.X, self.y, xs, separate_samples=True)


def _worker(rank, world, port, S, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from gpyreg_amd import sharding
    from oracle import gp_oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, X, y, hyp = orc.synthetic_problem(2, N=60, D=2, S=S)
        gp = _OracleGP(model, X, y)
        nlz, dnlz = sharding.nll_batch_sharded(gp, hyp, compute_grad=True)
        nlz0, none = sharding.nll_batch_sharded(gp, hyp, compute_grad=False)
        lo, hi = sharding.shard_bounds(S, rank, world)
        gp.posts = orc.posteriors(model, hyp[lo:hi], X, y, None)
        mu, s2 = sharding.predict_sharded(gp, X[:5])
        q.put((rank, nlz, dnlz, nlz0, none is None, gp.calls, mu, s2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [5, 4])
def test_sharded_nll_and_predict_world2(S):
    import torch.multiprocessing as mp

    from gpyreg_amd import sharding
    from oracle import gp_oracle as orc

    assert [sharding.shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sharding.shard_bounds(16, r, 8) for r in range(8)][-1] == (14, 16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    model, X, y, hyp = orc.synthetic_problem(2, N=60, D=2, S=S)
    ref = [orc.core(model, h, X, y, None, 1, 1) for h in hyp]
    ref_n = np.array([r[0] for r in ref])
    ref_d = np.stack([r[1] for r in ref])
    posts = orc.posteriors(model, hyp, X, y, None)
    rmu, rs2 = orc.predict(model, posts, X, y, X[:5], separate_samples=True)
    for rank, nlz, dnlz, nlz0, none_ok, calls, mu, s2 in res:
        assert np.array_equal(nlz, ref_n) and np.array_equal(dnlz, ref_d)  # every rank has it all
        assert np.array_equal(nlz0, ref_n) and none_ok
        lo, hi = sharding.shard_bounds(S, rank, 2)
        assert calls == [hi - lo, hi - lo]  # each rank evaluated only its own block
        assert np.array_equal(mu, rmu) and np.array_equal(s2, rs2)


def test_single_process_passthrough():
    from gpyreg_amd import sharding
    from oracle import gp_oracle as orc

    model, X, y, hyp = orc.synthetic_problem(2, N=30, D=2, S=3)
    gp = _OracleGP(model, X, y)
    nlz, dnlz = sharding.nll_batch_sharded(gp, hyp, compute_grad=True)
    assert gp.calls == [3] and nlz.shape == (3,) and dnlz.shape == (3, 5)


def _err_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from gpyreg_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        # (1) rank 1's block raises: both ranks must come back with ShardError (no hang in the collective)
        def boom(lo, hi):
            if rank == 1:
                raise FloatingPointError("singular block")
            return np.ones((hi - lo, 3)), np.zeros(hi - lo, bool)
        try:
            sharding.gather_rows(5, 3, boom)
            res.append("no error")
        except sharding.ShardError as e:
            res.append("ShardError")
        # (2) a per-sample failure flag (not an exception) reaches every rank
        rows, bad = sharding.gather_rows(
            5, 2, lambda lo, hi: (np.arange(lo, hi)[:, None] * np.ones((1, 2)), np.arange(lo, hi) == 4))
        res.append((rows[:, 0].tolist(), bad.tolist()))
        # (3) more ranks than samples: rank 1 has an empty block and still takes part
        rows, bad = sharding.gather_rows(1, 1, lambda lo, hi: (np.full((hi - lo, 1), 7.0), np.zeros(hi - lo, bool)))
        res.append(rows.ravel().tolist())
        # (4) unsynchronised callers: each rank drew its batch from its own RNG (different seeds), and (5) batches of
        # different LENGTH (the data exchange would have mismatched buffer shapes): ShardError on every rank BEFORE
        # the data exchange, and the group stays usable (6)
        mine = np.random.default_rng(100 + rank).standard_normal((5, 3))
        ident = lambda lo, hi: (mine[lo:hi], np.zeros(hi - lo, bool))
        for S_r, tok in ((5, sharding.fingerprint(mine)), (5 + rank, sharding.fingerprint(np.zeros(3)))):
            try:
                sharding.gather_rows(S_r, 3, (lambda lo, hi: (np.zeros((hi - lo, 3)), np.zeros(hi - lo, bool)))
                                     if S_r != 5 else ident, token=tok)
                res.append("no error")
            except sharding.ShardError as e:
                res.append("different batches" in str(e))
        same = np.random.default_rng(7).standard_normal((5, 3))
        rows, _ = sharding.gather_rows(5, 3, lambda lo, hi: (same[lo:hi], np.zeros(hi - lo, bool)),
                                       token=sharding.fingerprint(same))
        res.append(bool(np.array_equal(rows, same)))
        # (7) a result too wide for the frame (predictions at many points): header-only frame first, rows in a second
        # gather; an error and a per-sample flag travel there too
        wide = np.random.default_rng(8).standard_normal((5, 700))
        rows, bad = sharding.gather_rows(5, 700, lambda lo, hi: (wide[lo:hi], np.arange(lo, hi) == 3),
                                         token=sharding.fingerprint(wide))
        res.append(bool(np.array_equal(rows, wide)) and bad.tolist() == [False, False, False, True, False])
        try:
            sharding.gather_rows(5, 700, boom)
            res.append("no error")
        except sharding.ShardError:
            res.append("ShardError")
        # (8) the ranks disagree about WHICH of the two forms applies (rank 0: 3 columns, in the frame; rank 1: 700):
        # the first collective has the same shape either way, so both learn of it and neither hangs
        nc = 3 if rank == 0 else 700
        try:
            sharding.gather_rows(5, nc, lambda lo, hi: (np.zeros((hi - lo, nc)), np.zeros(hi - lo, bool)))
            res.append("no error")
        except sharding.ShardError as e:
            res.append("different batches" in str(e))
        # (9) the buffer pool is bounded: a block beyond _POOL_MAX_BYTES per rank is never kept, and many
        # distinct shapes (a loop over varying numbers of query points) stay inside the byte budget, oldest evicted
        before = sharding.pool_bytes()
        big = sharding._POOL_MAX_BYTES // 8 + 8
        w = np.ones((2, big))
        sharding.gather_rows(2, big, lambda lo, hi: (w[lo:hi], np.zeros(hi - lo, bool)))
        ok = sharding.pool_bytes() == before
        for M in range(60000, 60000 + 80):  # ~0.5 MB per rank each, 3 x that per set: 80 sets would be ~115 MB
            w = np.ones((2, M))
            sharding.gather_rows(2, M, lambda lo, hi, w=w: (w[lo:hi], np.zeros(hi - lo, bool)))
        res.append(ok and before < sharding.pool_bytes() <= sharding._POOL_BUDGET)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_a_failing_shard_raises_on_every_rank_instead_of_hanging():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_err_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert res[rank][0] == "ShardError"
        assert res[rank][1] == ([0.0, 1.0, 2.0, 3.0, 4.0], [False, False, False, False, True])
        assert res[rank][2] == [7.0]
        assert res[rank][3:6] == [True, True, True]
        assert res[rank][6:] == [True, "ShardError", True, True]


def _many_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from gpyreg_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for S in (2, 7, 16, 13):  # fewer samples than ranks, uneven blocks, bench.py's 2 per rank, a prime
            table = np.random.default_rng(S).standard_normal((S, 4))
            seen = []

            def block(lo, hi, table=table, seen=seen):
                seen.append((lo, hi))
                return table[lo:hi], np.arange(lo, hi) % 5 == 1

            rows, bad = sharding.gather_rows(S, 4, block, token=sharding.fingerprint(table))
            lo, hi = sharding.shard_bounds(S, rank, world)
            res[S] = (bool(np.array_equal(rows, table)), bool(np.array_equal(bad, np.arange(S) % 5 == 1)),
                      seen == ([(lo, hi)] if hi > lo else []))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_uneven_and_empty_blocks():
    """The 8-GPU layout on the CPU (gloo): blocks of unequal size, ranks with NO sample (S < world) -- the rows come
    back complete and in order on every rank, each rank computes its own block once (or nothing)."""
    import torch.multiprocessing as mp

    from gpyreg_amd import sharding

    world = 8
    assert [sharding.shard_bounds(2, r, world) for r in range(world)][:3] == [(0, 1), (1, 2), (2, 2)]
    assert sum(hi - lo for lo, hi in (sharding.shard_bounds(13, r, world) for r in range(world))) == 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        for S in (2, 7, 16, 13):
            assert res[rank][S] == (True, True, True), (rank, S, res[rank][S])


def _rccl_agreement_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import warnings

    import torch.distributed as dist

    from gpyreg_amd import _rccl, sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        made = []

        class FakeComm:
            """Stands in for _rccl.Comm (which needs RCCL and a GPU): fails where `fail` says, records what happened."""
            fail = {}

            def __init__(self, group):
                made.append(self)
                self.connected = self.destroyed = False
                if FakeComm.fail.get("init") == rank:
                    raise RuntimeError("librccl.so: cannot open shared object file")

            def connect(self, group):
                if FakeComm.fail.get("connect") == rank:
                    raise RuntimeError("ncclCommInitRank: unhandled system error")
                self.connected = True

            def destroy(self):
                self.destroyed = True

        _rccl.Comm = FakeComm
        res = []
        warnings.simplefilter("ignore")
        for case, env_on, fail in [("init fails on rank 1", (True, True), {"init": 1}),
                                   ("connect fails on rank 0", (True, True), {"connect": 0}),
                                   ("only rank 0 asks for it", (True, False), {}),
                                   ("nobody asks", (False, False), {}),
                                   ("everything works", (True, True), {})]:
            os.environ["GPYREG_AMD_EXCHANGE"] = "rccl" if env_on[rank] else "torch"
            _rccl._state["failed"] = False
            FakeComm.fail = fail
            del made[:]
            g = dist.new_group([0, 1])
            got = _rccl.comm_for(g)
            again = _rccl.comm_for(g)  # decided once per group: no second agreement (it would need the peer)
            res.append((case, got is not None, again is got, [m.destroyed for m in made]))
            # whatever was decided, the exchange itself works on this group afterwards
            rows, _ = sharding.gather_rows(4, 2, lambda lo, hi: (np.arange(lo, hi)[:, None] * np.ones((1, 2)),
                                                                 np.zeros(hi - lo, bool)), group=g)
            assert rows[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0]
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_direct_exchange_falls_back_on_every_rank_or_none():
    """gpyreg_amd/_rccl.py: whether a process group takes the direct RCCL exchange is agreed by ALL its ranks (all-reduce
    MIN of a success flag after every stage of the set-up).  With a failure forced on ONE rank -- or the option set on one
    rank only -- BOTH ranks must come back with "torch path" (a rank that fell back alone would leave its peer waiting in
    ncclAllGather: VERDICT r5 item 6), a half-made communicator is destroyed, and when nothing fails both take it."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_agreement_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        by_case = {c: (direct, cached, destroyed) for c, direct, cached, destroyed in res[rank]}
        for c in ("init fails on rank 1", "connect fails on rank 0", "only rank 0 asks for it", "nobody asks"):
            assert by_case[c][0] is False and by_case[c][1], (rank, c)  # the torch path on EVERY rank
        assert by_case["everything works"][0] is True and by_case["everything works"][1]
        assert by_case["connect fails on rank 0"][2] == [True]  # the half-made communicator was destroyed on both ranks
        assert by_case["everything works"][2] == [False]
    # "init fails on rank 1": rank 0 had made its half (stage one worked there) and destroyed it; rank 1's raised in __init__
    assert res[0][0][0] == "init fails on rank 1" and res[0][0][3] == [True] and res[1][0][3] == [False]

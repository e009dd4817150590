"""Sharding of independent hyperparameter samples over the GPUs of one node.

Every caller of the hot path that holds more than one hyperparameter vector is a loop
with no cross-iteration dependence in the reference (f_min_fill.py:174-176,
gaussian_process.py:876-879, :1177-1187, :1727): the unit of work is ONE hyperparameter
vector.  One process per GPU (torchrun); X and y are replicated (a few MB); the rows of
``hyp`` are block-partitioned over ranks; no K/L data ever crosses xGMI.  The single
exchange step is an all-gather of the per-sample result vectors [nlZ | dnlZ] (or of the
predictive moments): a few hundred bytes per rank, latency bound, one RCCL call.
"""

from __future__ import annotations

import numpy as np


# Time this process has spent in the exchange step of sharded calls (bench.py reports it per step): from the end of
# the rank's own block to the moment the gathered rows are back on the host -- waiting for the slowest rank, the
# agreement and data all-gathers and their host <-> device copies.
_stats = {"seconds": 0.0, "calls": 0}


def reset_stats():
    _stats["seconds"], _stats["calls"] = 0.0, 0


def stats():
    return dict(_stats)


def shard_bounds(S: int, rank: int, world: int):
    """Rows [lo, hi) of rank ``rank``: contiguous blocks, sizes differ by at most one."""
    base, rem = divmod(S, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist


# Buffers of the exchanges, reused from call to call: a gather is a few hundred bytes per rank, so what it costs is the
# host work around the collective (tools/exchange_probe.py: 59 us to issue and 90-120 us to complete one with freshly
# allocated tensors, a pageable upload and `.cpu()`; two per sharded call).  Per (rows, columns, world, device): the
# device block and result, and pinned host images of both (asynchronous copies either way).  A set is taken for the
# life of one _Gather and handed back by `result`, so two gathers in flight never share one.
_POOL = {}


def _buffers(maxrows, C, world, dev):
    import torch

    key = (maxrows, C, world, str(dev))
    free = _POOL.setdefault(key, [])
    if free:
        return key, free.pop()
    cuda = dev.type == "cuda"
    buf = torch.zeros((maxrows, C), dtype=torch.float64, device=dev)
    out = torch.empty((world * maxrows, C), dtype=torch.float64, device=dev)
    hin = torch.zeros((maxrows, C), dtype=torch.float64, pin_memory=cuda) if cuda else None
    hout = torch.empty((world * maxrows, C), dtype=torch.float64, pin_memory=cuda) if cuda else None
    return key, (buf, out, hin, hout)


class _Gather:
    """An all-gather of row blocks in flight: the constructor issues it (asynchronously), ``result`` waits and returns
    the full (S, C) array.  RCCL over xGMI when the backend is nccl."""

    def __init__(self, local: np.ndarray, S: int, group=None):
        import torch

        dist = _dist()
        self.S, self.group = S, group
        self.world = dist.get_world_size(group)
        C = local.shape[1]
        self.maxrows = -(-S // self.world)
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.key, (buf, out, hin, hout) = _buffers(self.maxrows, C, self.world, dev)
        n = local.shape[0]
        src = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
        if hin is not None:  # device group: through the pinned image, asynchronously on the current stream
            hin.zero_()
            hin[:n] = src
            buf.copy_(hin, non_blocking=True)
        else:
            buf.zero_()
            buf[:n] = src
        self.bufs = (buf, out, hin, hout)  # held until the collective has completed
        self.work = dist.all_gather_into_tensor(out, buf, group=group, async_op=True)

    def result(self) -> np.ndarray:
        import torch

        buf, out, hin, hout = self.bufs
        self.work.wait()
        if hout is not None:
            hout.copy_(out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            full = hout.numpy().reshape(self.world, self.maxrows, -1)
        else:
            full = out.numpy().reshape(self.world, self.maxrows, -1)
        rows = []
        for r in range(self.world):
            lo, hi = shard_bounds(self.S, r, self.world)
            rows.append(full[r, : hi - lo])
        res = np.concatenate(rows, axis=0)  # (a copy: the buffers go back to the pool)
        _POOL[self.key].append(self.bufs)
        self.bufs = None
        return res


def _all_gather_rows(local: np.ndarray, S: int, group=None) -> np.ndarray:
    """Gather the row blocks of every rank into the full (S, C) array on every rank."""
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return local
    return _Gather(local, S, group).result()


def active_group(group=None):
    """``(rank, world)`` when a torch.distributed process group with more than one rank is
    initialised (torch is only touched if the caller imported it already), else None."""
    import sys

    if "torch" not in sys.modules:
        return None
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return None
    return dist.get_rank(group), dist.get_world_size(group)


class ShardError(RuntimeError):
    """A rank failed inside a sharded evaluation (raised on EVERY rank after the exchange)."""


def fingerprint(*arrays) -> float:
    """CRC-32 of the bytes (and shapes) of the arguments every rank is supposed to pass identically; exact in a
    float64, so it can ride in the status row of the exchange."""
    import zlib

    crc = 0
    for a in arrays:
        a = np.ascontiguousarray(a)
        crc = zlib.crc32(repr((a.shape, a.dtype.str)).encode(), crc)
        crc = zlib.crc32(a.tobytes(), crc)
    return float(crc)


def gather_rows(S: int, ncols: int, compute_local, group=None, token: float = 0.0):
    """Run ``compute_local(lo, hi) -> (rows (hi-lo, ncols), bad (hi-lo,) bool)`` on this rank's block
    and return the full ``(S, ncols)`` array and the full ``bad`` mask on every rank.

    A rank whose block raises does NOT leave the others waiting in the collective: the exception is
    caught, the rank still enters the exchange with a status row, and every rank raises
    ``ShardError`` afterwards (a non-positive-definite sample on one shard is an expected event
    during fitting; it must not become a hang).

    The ranks must have been called with the SAME batch.  An agreement row (fixed shape, exchanged
    asynchronously UNDER the local computation) carries S, ncols and ``token`` -- the caller's
    ``fingerprint`` of the full argument arrays: if they differ between ranks (unsynchronised RNG seeds
    in ``fit``, a speculative batch of another length) every rank raises ``ShardError`` before the data
    exchange, whose buffer shapes would disagree -- instead of silently stitching together rows of
    different batches.  One blocking collective per call: the data exchange, which also carries a status
    row per rank."""
    rw = active_group(group)
    if rw is None:
        rows, bad = compute_local(0, S)
        return np.asarray(rows, dtype=float).reshape(S, ncols), np.asarray(bad, dtype=bool)
    rank, world = rw
    lo, hi = shard_bounds(S, rank, world)
    # 1. the agreement exchange -- one fixed-shape row per rank: [S, ncols, fingerprint of the arguments] -- is issued
    #    BEFORE the local computation and completes under it
    agreement = _Gather(np.array([[float(S), float(ncols), float(token)]]), world, group)
    # 2. this rank's block
    local = np.zeros((hi - lo, ncols + 1))
    err = None
    try:
        if hi > lo:
            rows, bad = compute_local(lo, hi)
            local[:, :ncols] = np.asarray(rows, dtype=float).reshape(hi - lo, ncols)
            local[:, ncols] = np.asarray(bad, dtype=float)
    except Exception as e:  # noqa: BLE001 - exchanged, then raised on every rank
        err = e
        local[:] = 0.0
    # 3. every rank sees the same agreement rows and takes the same branch
    import time

    t_exchange = time.perf_counter()
    seen = agreement.result()
    if np.any(seen != seen[0]):
        detail = ", ".join(f"rank {r}: S={int(seen[r, 0])} cols={int(seen[r, 1])} crc={int(seen[r, 2]):08x}"
                           for r in range(world))
        raise ShardError(
            "the ranks of a sharded evaluation were called with different batches (" + detail + "). Every rank "
            "must pass the same hyperparameter rows: seed NumPy's global RNG identically on all ranks before "
            "GP.fit, or set gp.shard = False to keep this GP rank-local") from err
    # 4. the data exchange (shapes agree now): the rank's rows plus ONE status row, so that a rank with an empty block
    #    that failed is heard too.  Blocks are padded to the same height; the status row rides at index maxrows.
    maxrows = -(-S // world)
    block = np.zeros((maxrows + 1, ncols + 1))
    block[: hi - lo] = local
    block[maxrows, ncols] = 0.0 if err is None else 1.0
    gathered = _Gather(block, world * (maxrows + 1), group).result().reshape(world, maxrows + 1, ncols + 1)
    _stats["seconds"] += time.perf_counter() - t_exchange
    _stats["calls"] += 1
    if err is not None:
        raise ShardError(f"rank {rank}: {type(err).__name__}: {err}") from err
    failed = [r for r in range(world) if gathered[r, maxrows, ncols] != 0.0]
    if failed:
        raise ShardError(f"sharded evaluation failed on rank(s) {failed}")
    full = np.concatenate([gathered[r, : shard_bounds(S, r, world)[1] - shard_bounds(S, r, world)[0]]
                           for r in range(world)], axis=0)
    return full[:, :ncols].copy(), full[:, ncols] != 0.0


def nll_batch_sharded(gp, hyp: np.ndarray, compute_grad: bool = False, group=None):
    """``gp.nll_batch`` with the samples sharded over the process group.

    Every rank passes the SAME ``hyp`` (S, hyp_N), evaluates only its block on its own
    GPU and receives the full result: nlZ (S,), dnlZ (S, hyp_N) | None.
    """
    hyp = np.atleast_2d(np.asarray(hyp, dtype=float))
    S, hyp_N = hyp.shape
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1 or getattr(gp, "shard", False):
        return gp.nll_batch(hyp, compute_grad)  # a gpyreg_amd.GP shards by itself
    lo, hi = shard_bounds(S, dist.get_rank(group), dist.get_world_size(group))
    C = 1 + (hyp_N if compute_grad else 0)
    local = np.zeros((hi - lo, C))
    if hi > lo:
        nlz, dnlz = gp.nll_batch(hyp[lo:hi], compute_grad)
        local[:, 0] = nlz
        if compute_grad:
            local[:, 1:] = dnlz
    full = _all_gather_rows(local, S, group)
    return full[:, 0].copy(), (full[:, 1:].copy() if compute_grad else None)


def predict_sharded(gp, x_star: np.ndarray, group=None):
    """Per-sample predictive mean / variance with the posterior samples sharded: each
    rank's ``gp`` holds the posteriors of ITS block of hyperparameter samples; returns
    mu, s2 of shape (M, S_total) on every rank (``separate_samples=True`` semantics of
    gaussian_process.py:1663-1787; sample averaging is then rank-local arithmetic)."""
    mu, s2 = gp.predict(x_star, separate_samples=True)
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1 or getattr(gp, "shard", False):
        return mu, s2
    import torch

    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64)
    counts[dist.get_rank(group)] = mu.shape[1]
    if dist.get_backend(group) == "nccl":
        counts = counts.cuda()
    dist.all_reduce(counts, group=group)
    S = int(counts.sum().item())
    # rows = samples so that the row gather applies
    loc = np.concatenate([mu.T, s2.T], axis=1)
    sizes = [int(c) for c in counts.cpu().tolist()]
    if any(sizes[r] != shard_bounds(S, r, world)[1] - shard_bounds(S, r, world)[0] for r in range(world)):
        raise ValueError("posterior samples must be block-partitioned with shard_bounds()")
    full = _all_gather_rows(loc, S, group)
    M = mu.shape[0]
    return full[:, :M].T.copy(), full[:, M:].T.copy()

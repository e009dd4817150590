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
# host work around the collective (tools/exchange_probe.py, profiles/r04_exchange_probe.txt: 59 us to issue and 90-120 us
# to complete one with freshly allocated tensors, a pageable upload and `.cpu()`).  Per (rows, columns, world, device):
# the device block and result, and pinned host images of both (asynchronous copies either way).  A set is taken for the
# life of one _Gather and handed back when it ends -- by `result`, also when the wait raises -- so two gathers in flight
# never share one.  The pool is BOUNDED: blocks beyond _POOL_MAX_BYTES per rank are never kept, and the free
# sets together hold at most _POOL_BUDGET bytes -- the least recently used shapes go first -- so a loop over varying
# numbers of query points (predict / predict_full / quad gather M columns) cannot grow pinned memory without bound.
_POOL = {}  # key -> list of free buffer sets; dict order = recency of use
_POOL_MAX_BYTES = 4 << 20    # per-rank block size up to which a shape is pooled (a pinned allocation costs ~60 us)
_POOL_BUDGET = 64 << 20      # bytes of device (and as many pinned host) memory the free sets may hold
_pool_bytes = [0]


def _set_bytes(key):
    maxrows, C, world, _ = key
    return (world + 1) * maxrows * C * 8


def _buffers(maxrows, C, world, dev):
    import torch

    key = (maxrows, C, world, str(dev))
    free = _POOL.get(key)
    if free:
        _POOL[key] = _POOL.pop(key)  # most recently used
        _pool_bytes[0] -= _set_bytes(key)
        return key, free.pop()
    cuda = dev.type == "cuda"
    buf = torch.empty((maxrows, C), dtype=torch.float64, device=dev)  # (every exchange writes the whole block)
    out = torch.empty((world * maxrows, C), dtype=torch.float64, device=dev)
    hin = torch.zeros((maxrows, C), dtype=torch.float64, pin_memory=cuda) if cuda else None
    hout = torch.empty((world * maxrows, C), dtype=torch.float64, pin_memory=cuda) if cuda else None
    # NumPy views of the host images (and of the CPU tensors of a gloo group): filling and reading them costs a
    # memcpy, not a torch indexing call each
    vin = (hin if cuda else buf).numpy()
    vout = (hout if cuda else out).numpy()
    return key, (buf, out, hin, hout, vin, vout)


def _give_back(key, bufs):
    if key[0] * key[1] * 8 > _POOL_MAX_BYTES:
        return  # a large, caller-shaped gather: not kept
    _POOL.setdefault(key, []).append(bufs)
    _POOL[key] = _POOL.pop(key)
    _pool_bytes[0] += _set_bytes(key)
    while _pool_bytes[0] > _POOL_BUDGET and _POOL:
        old = next(iter(_POOL))
        if _POOL[old]:
            _POOL[old].pop()
            _pool_bytes[0] -= _set_bytes(old)
        if not _POOL[old]:
            del _POOL[old]


def pool_bytes() -> int:
    """Bytes of device memory (and as many of pinned host memory) held by the free buffer sets of the exchanges."""
    return _pool_bytes[0]


class _Gather:
    """An all-gather of row blocks in flight: the constructor issues it (asynchronously), ``result`` waits and returns
    the full (S, C) array.  RCCL over xGMI when the backend is nccl.  ``raw=True``: ``result`` returns the
    (world, maxrows, C) block as gathered (a copy), without cutting the rows of each rank to its share."""

    def __init__(self, local: np.ndarray, S: int, group=None, raw: bool = False):
        import torch

        dist = _dist()
        self.S, self.group, self.raw = S, group, raw
        self.world = dist.get_world_size(group)
        C = local.shape[1]
        self.maxrows = -(-S // self.world)
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.key, self.bufs = _buffers(self.maxrows, C, self.world, dev)  # held until the collective has completed
        buf, out, hin, hout, vin, vout = self.bufs
        self.direct = None
        try:
            n = local.shape[0]
            vin[:n] = local
            if n < self.maxrows:
                vin[n:] = 0.0
            if hin is not None:
                from . import _rccl

                self.direct = _rccl.comm_for(group)  # GPYREG_AMD_EXCHANGE=rccl: one direct RCCL call, no torch collective
            if self.direct is not None:
                self.direct.issue(hin, buf, out, hout)
                self.work = None
                return
            if hin is not None:  # device group: through the pinned image, asynchronously on the current stream
                buf.copy_(hin, non_blocking=True)
            self.work = dist.all_gather_into_tensor(out, buf, group=group, async_op=True)
        except BaseException:
            self.bufs = None  # (not handed back: their state is unknown)
            raise

    def result(self) -> np.ndarray:
        import torch

        buf, out, hin, hout, vin, vout = self.bufs
        try:
            if self.direct is not None:
                self.direct.wait()  # upload, all-gather and download were enqueued together
            else:
                self.work.wait()
                if hout is not None:
                    hout.copy_(out, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
            full = vout.reshape(self.world, self.maxrows, -1)
            if self.raw:
                res = full.copy()
            else:
                rows = []
                for r in range(self.world):
                    lo, hi = shard_bounds(self.S, r, self.world)
                    rows.append(full[r, : hi - lo])
                res = np.concatenate(rows, axis=0)  # (a copy: the buffers go back to the pool)
        except BaseException:
            # a wait that raised (time-out, asynchronous RCCL error) says nothing about the collective having stopped
            # writing out / hout: the buffers are dropped, as __init__ drops them, never handed to the next gather
            self.bufs = None
            raise
        _give_back(self.key, self.bufs)  # completed: nothing is in flight on these buffers any more
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


# Every sharded call opens with ONE gather of a fixed shape, the frame: FRAME doubles per rank whatever the batch --
# [S, ncols, token, status, payload ...].  A fixed shape cannot disagree between ranks, so no separate agreement
# exchange has to come first: results of up to FRAME - 4 doubles per rank (cfg3: 2 rows x 15, cfg5: 8 x 13, the
# 1024-point design of fit on 8 GPUs: 128 x 2) ride in the frame itself, after the local computation -- one collective
# per call (round 4: two; profiles/r05_exchange_probe.txt).  Larger results (predictions at many points) send the frame
# BEFORE the local computation, header only, so that it completes under it, and their rows in a second gather once the
# shapes are known to agree.
FRAME = 512
_HDR = 4


def _disagreement(seen, world, err):
    detail = ", ".join(f"rank {r}: S={int(seen[r, 0])} cols={int(seen[r, 1])} crc={int(seen[r, 2]):08x}"
                       for r in range(world))
    return ShardError(
        "the ranks of a sharded evaluation were called with different batches (" + detail + "). Every rank "
        "must pass the same hyperparameter rows: seed NumPy's global RNG identically on all ranks before "
        "GP.fit, or set gp.shard = False to keep this GP rank-local")


def gather_rows(S: int, ncols: int, compute_local, group=None, token: float = 0.0):
    """Run ``compute_local(lo, hi) -> (rows (hi-lo, ncols), bad (hi-lo,) bool)`` on this rank's block
    and return the full ``(S, ncols)`` array and the full ``bad`` mask on every rank.

    A rank whose block raises does NOT leave the others waiting in the collective: the exception is
    caught, the rank still enters the exchange with its status set, and every rank raises
    ``ShardError`` afterwards (a non-positive-definite sample on one shard is an expected event
    during fitting; it must not become a hang).

    The ranks must have been called with the SAME batch.  The frame (fixed shape, see FRAME) carries S,
    ncols and ``token`` -- the caller's ``fingerprint`` of the full argument arrays: if they differ
    between ranks (unsynchronised RNG seeds in ``fit``, a speculative batch of another length) every rank
    raises ``ShardError`` -- before any exchange whose buffer shapes would disagree -- instead of silently
    stitching together rows of different batches."""
    import time

    rw = active_group(group)
    if rw is None:
        rows, bad = compute_local(0, S)
        return np.asarray(rows, dtype=float).reshape(S, ncols), np.asarray(bad, dtype=bool)
    rank, world = rw
    lo, hi = shard_bounds(S, rank, world)
    maxrows = -(-S // world)
    header = np.array([float(S), float(ncols), float(token), 0.0])
    in_frame = maxrows * (ncols + 1) <= FRAME - _HDR

    def run_local():
        local = np.zeros((hi - lo, ncols + 1))
        try:
            if hi > lo:
                rows, bad = compute_local(lo, hi)
                local[:, :ncols] = np.asarray(rows, dtype=float).reshape(hi - lo, ncols)
                local[:, ncols] = np.asarray(bad, dtype=float)
            return local, None
        except Exception as e:  # noqa: BLE001 - exchanged, then raised on every rank
            local[:] = 0.0
            return local, e

    def stitch(block_of):
        full = np.concatenate([block_of(r)[: shard_bounds(S, r, world)[1] - shard_bounds(S, r, world)[0]]
                               for r in range(world)], axis=0)
        return full[:, :ncols].copy(), full[:, ncols] != 0.0

    frame = np.zeros((1, FRAME))
    frame[0, :_HDR] = header
    if in_frame:
        # ONE collective: this rank's rows and its status ride in the frame, after the local computation
        local, err = run_local()
        t_exchange = time.perf_counter()
        frame[0, 3] = 0.0 if err is None else 1.0
        frame[0, _HDR:_HDR + local.size] = local.ravel()
        seen = _Gather(frame, world, group, raw=True).result().reshape(world, FRAME)
        _stats["seconds"] += time.perf_counter() - t_exchange
        _stats["calls"] += 1
        if np.any(seen[:, :3] != seen[0, :3]):
            raise _disagreement(seen, world, err) from err
        status = seen[:, 3]
        block_of = lambda r: seen[r, _HDR:_HDR + maxrows * (ncols + 1)].reshape(maxrows, ncols + 1)  # noqa: E731
    else:
        # the frame (header only) is issued BEFORE the local computation and completes under it; the rows follow in a
        # second gather once every rank is known to hold the same shapes
        agreement = _Gather(frame, world, group, raw=True)
        local, err = run_local()
        t_exchange = time.perf_counter()
        seen = agreement.result().reshape(world, FRAME)
        if np.any(seen[:, :3] != seen[0, :3]):
            raise _disagreement(seen, world, err) from err
        # blocks are padded to the same height; ONE status row per rank rides at index maxrows, so that a rank with an
        # empty block that failed is heard too
        block = np.zeros((maxrows + 1, ncols + 1))
        block[: hi - lo] = local
        block[maxrows, ncols] = 0.0 if err is None else 1.0
        gathered = _Gather(block, world * (maxrows + 1), group, raw=True).result()
        _stats["seconds"] += time.perf_counter() - t_exchange
        _stats["calls"] += 1
        status = gathered[:, maxrows, ncols]
        block_of = lambda r: gathered[r]  # noqa: E731
    if err is not None:
        raise ShardError(f"rank {rank}: {type(err).__name__}: {err}") from err
    failed = [r for r in range(world) if status[r] != 0.0]
    if failed:
        raise ShardError(f"sharded evaluation failed on rank(s) {failed}")
    return stitch(block_of)


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

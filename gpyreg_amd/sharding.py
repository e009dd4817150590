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


def _all_gather_rows(local: np.ndarray, S: int, group=None) -> np.ndarray:
    """Gather the row blocks of every rank into the full (S, C) array on every rank."""
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return local
    import torch

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C = local.shape[1]
    maxrows = -(-S // world)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((maxrows, C), dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    out = torch.empty((world * maxrows, C), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)  # RCCL over xGMI when backend == nccl
    out = out.cpu().numpy().reshape(world, maxrows, C)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(S, r, world)
        rows.append(out[r, : hi - lo])
    assert rank < world
    return np.concatenate(rows, axis=0)


def nll_batch_sharded(gp, hyp: np.ndarray, compute_grad: bool = False, group=None):
    """``gp.nll_batch`` with the samples sharded over the process group.

    Every rank passes the SAME ``hyp`` (S, hyp_N), evaluates only its block on its own
    GPU and receives the full result: nlZ (S,), dnlZ (S, hyp_N) | None.
    """
    hyp = np.atleast_2d(np.asarray(hyp, dtype=float))
    S, hyp_N = hyp.shape
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return gp.nll_batch(hyp, compute_grad)
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
    if dist is None or dist.get_world_size(group) == 1:
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

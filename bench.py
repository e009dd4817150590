#!/usr/bin/env python3
"""bench.py -- GP fit-evaluations/sec (NLL + gradient) on MI355X, at 1/2/4/8 GPUs.

A "step" is one batched pass of the hot path: `GP.nll_batch(hyp[S], compute_grad=True)`
for S hyperparameter samples at the workload BASELINE.json's metric is quoted on
(N=4096, D=10, Matern-5/2 ARD, ConstantMean, constant Gaussian noise, fp64).
X and y are resident in HBM before the timed region; the per-step H2D of the
hyperparameter-derived vectors and D2H of (nlZ, dnlZ) are inside it.

    python bench.py --gpus N --steps K --warmup W            # starts the N ranks itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU; hyperparameter samples are the independent units (the loops of
f_min_fill.py:174-176, gaussian_process.py:876-879, :1727).  Every rank passes the SAME global batch
to `GP.nll_batch`, which block-partitions its rows over the process group (gpyreg_amd/sharding.py),
evaluates its block on its own GPU and all-gathers the per-sample [nlZ | dnlZ] rows over RCCL -- the
one exchange step of the path, done every step, inside the product's own API.

    --scaling config   (default) the global batch is BASELINE.json's: cfg3 = 16 samples over the N
                       GPUs (2 per GPU at 8), cfg5 = 64 (8 per GPU).  Total work fixed: "strong".
    --scaling weak     every GPU gets the configuration's S samples (N x S in all): "weak".

When `--gpus N` > 1 and no launcher has set RANK / WORLD_SIZE, this process starts the N ranks as
fresh child processes (before anything touches the GPU: the parent never imports torch), hands rank
0's JSON line through and exits non-zero if any rank fails.

    --mode fit         NLL + gradient (the headline metric)
    --mode nll         NLL only (the 1024-point design of `fit` and the slice sampler)
    --mode predict     GP.predict at M query points for the S posteriors of the configuration
                       (gaussian_process.py:1741-1764): point-samples/s

Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    2: dict(N=2048, D=5, kernel="se", degree=0, S=1),
    3: dict(N=4096, D=10, kernel="matern", degree=5, S=16),
    4: dict(N=16384, D=20, kernel="rq", degree=0, S=1),
    5: dict(N=8192, D=8, kernel="se", degree=0, S=64),
    # (not a BASELINE configuration: twice the N that the device library accepted before round 6 -- the reference
    # factorizes whatever fits host memory, gaussian_process.py:2415-2417, :2477-2484)
    6: dict(N=32768, D=5, kernel="se", degree=0, S=2),
}
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet, dense fp64 matrix
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table


def synthetic_problem(cfg_idx, S, seed_shift=0):
    """Seeded synthetic inputs of SURVEY.md 8(d): draw order X, y-noise, then hyp."""
    c = CONFIGS[cfg_idx]
    N, D = c["N"], c["D"]
    rng = np.random.default_rng(1000 + cfg_idx)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
    base = [np.log(1.5 * np.sqrt(D) * (1 + 0.1 * d / D)) for d in range(D)] + [0.0]
    if c["kernel"] == "rq":
        base.append(0.0)
    base += [np.log(0.1), 0.0]
    base = np.asarray(base)
    if seed_shift:
        rng = np.random.default_rng(1000 + cfg_idx + 7919 * seed_shift)
    hyp = base + 0.1 * rng.standard_normal((S, base.size))
    return X, y, hyp


def make_gp(cfg_idx, dtype):
    import gpyreg_amd as gpr

    c = CONFIGS[cfg_idx]
    cov = {
        "se": gpr.covariance_functions.SquaredExponential,
        "rq": gpr.covariance_functions.RationalQuadraticARD,
    }.get(c["kernel"])
    cov = gpr.covariance_functions.Matern(c["degree"]) if c["kernel"] == "matern" else cov()
    return gpr.GP(c["D"], cov, gpr.mean_functions.ConstantMean(),
                  gpr.noise_functions.GaussianNoise(constant_add=True), dtype=dtype)


def _oracle_model(cfg_idx):
    c = CONFIGS[cfg_idx]
    return dict(kernel=c["kernel"], degree=c["degree"], mean="const", noise=(1, 0, 0))


def _host_description():
    """CPU model and the numerical stack the CPU baseline ran on (SURVEY.md 8d asks for them)."""
    import platform

    import scipy

    model = platform.processor() or "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas = []
    try:
        from threadpoolctl import threadpool_info

        for lib in threadpool_info():
            blas.append({k: lib.get(k) for k in ("internal_api", "version", "num_threads", "threading_layer", "architecture")})
    except Exception:  # noqa: BLE001 - description only
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "python": platform.python_version(),
            "numpy": np.__version__, "scipy": scipy.__version__, "blas": blas}


def _blas_threads(host):
    # (the BLAS pools only: an OpenMP runtime that torch brought into the process takes no part in the oracle's work)
    return max([b.get("num_threads") or 1 for b in host["blas"]
                if b.get("internal_api") in ("openblas", "mkl", "blis", "flexiblas")] or [1])


def _oracle_eval_seconds(cfg_idx, repeats, warmup, grad=True):
    """Wall-clock seconds of ``repeats`` evaluations of sample 0 by the CPU oracle (after ``warmup``
    untimed ones), plus the last result.  Also the body of the 1-thread child process."""
    from oracle import gp_oracle as orc  # cpu_baseline leg only

    X, y, hyp = synthetic_problem(cfg_idx, 1)
    model = _oracle_model(cfg_idx)
    times, out = [], None
    for it in range(warmup + repeats):
        t0 = time.perf_counter()
        out = orc.core(model, hyp[0], X, y, None, 1, 1 if grad else 0)
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    return times, (out if grad else (out, None))


def cpu_baseline(cfg_idx, gpu_nlz0=None, gpu_dnlz0=None, grad=True):
    """The CPU oracle (NumPy/SciPy restatement pinned to the reference) timed on this host's
    cores on a BOUNDED sample of the same workload: sample 0 of the batch, one
    warm-up evaluation then the median of three (default BLAS threading = all cores), and one
    evaluation in a child process restricted to ONE BLAS thread.  The oracle's value and gradient
    double as a live parity check of the GPU result (``grad_rel_err``).  (BASELINE.md asks for 3 warm-ups
    and a median of >= 10: at 13 s per evaluation that does not fit the few minutes a default run may take;
    the protocol used is stated in ``sample``.)"""
    import subprocess

    c = CONFIGS[cfg_idx]
    # cfg2 (0.7 s per evaluation): BASELINE.md's own protocol, 3 warm-ups + median of 10.  cfg3 (14 s): 1 warm-up +
    # median of 3.  cfg5 (N=8192): one evaluation (minutes).  cfg4 (N=16384, RQ): the reference has no fp32 path and its
    # (N,N,22) gradient tensor needs 47 GB, so the CPU figure is one fp64 NLL-only evaluation; cfg6
    # (N=32768): likewise NLL only -- the (N,N,6) tensor is 52 GB on top of K, L and the inverse.
    full = cfg_idx in (2, 3)
    grad = grad and cfg_idx not in (4, 6)
    repeats, warmup = {2: (10, 3), 3: (3, 1)}.get(cfg_idx, (1, 0))
    times, (nlz, dnlz) = _oracle_eval_seconds(cfg_idx, repeats=repeats, warmup=warmup, grad=grad)
    med = float(np.median(times))
    if not grad:
        gpu_dnlz0 = None
    one = None
    try:
        if not full:
            raise RuntimeError("single-thread figure only for cfg2/cfg3")
        env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "t, _ = bench._oracle_eval_seconds(%d, 1, 0, %r); print(json.dumps(t))" % (ROOT, cfg_idx, grad))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        one = float(json.loads(r.stdout.strip().splitlines()[-1])[0])
    except Exception:  # noqa: BLE001 - the 1-thread figure is optional
        one = None
    host = _host_description()
    # `cores` = the threads the evaluation can actually use: the BLAS pool (NumPy's elementwise passes, which
    # dominate, are single-threaded); the logical CPU count is in host.logical_cpus
    out = dict(value=1.0 / med, unit="fit-evals/s" if grad else "NLL-evals/s", cores=_blas_threads(host), kind="port",
               sample=f"sample 0 of {c['S']}, {'NLL+grad' if grad else 'NLL only (fp64)'}, N={c['N']} D={c['D']} "
                      f"{c['kernel']}{c['degree'] or ''}: "
                      f"{('%d warm-up(s) + median of %d evaluations' % (warmup, repeats)) if repeats > 1 else '1 evaluation'} "
                      f"({', '.join('%.2f' % t for t in times)} s), default BLAS threading",
               seconds_per_eval=med,
               single_thread={"value": None if one is None else 1.0 / one, "seconds_per_eval": one,
                              "how": "child process, OPENBLAS/OMP/MKL_NUM_THREADS=1, one evaluation"},
               host=host, nlz=float(nlz))
    if gpu_nlz0 is not None:
        out["nlz_rel_err"] = float(abs(gpu_nlz0 - nlz) / max(1.0, abs(nlz)))
    if gpu_dnlz0 is not None:
        # per component, relative to that component (floor: 1e-3 of the largest one, below which a component is a
        # cancellation of terms a thousand times its size); and the looser figure relative to the largest component
        err = np.abs(gpu_dnlz0 - dnlz)
        out["grad_rel_err"] = float((err / np.maximum(np.abs(dnlz), 1e-3 * np.abs(dnlz).max())).max())
        out["grad_rel_err_vs_largest"] = float(err.max() / np.abs(dnlz).max())
    return out


def cpu_baseline_predict(cfg_idx, xs, gpu_mu0, gpu_s20):
    """`predict` of the CPU oracle for ONE posterior sample (sample 0) at the same query points: the posterior is
    built untimed, then one warm-up call and the median of three are timed.  Doubles as the live parity check."""
    from oracle import gp_oracle as orc  # cpu_baseline leg only

    c = CONFIGS[cfg_idx]
    X, y, hyp = synthetic_problem(cfg_idx, 1)
    model = _oracle_model(cfg_idx)
    posts = orc.posteriors(model, hyp[:1], X, y, None)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        mu, s2 = orc.predict(model, posts, X, y, xs, separate_samples=True)
        if it:
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    host = _host_description()
    M = xs.shape[0]
    return dict(value=M / med, unit="point-samples/s", cores=_blas_threads(host), kind="port",
                sample=f"posterior sample 0 of {c['S']}, M={M} query points, N={c['N']} D={c['D']} "
                       f"{c['kernel']}{c['degree'] or ''}: 1 warm-up + median of 3 predict calls "
                       f"({', '.join('%.3f' % t for t in times)} s), default BLAS threading; posterior built untimed",
                seconds_per_call=med, host=host,
                mu_abs_err=float(np.abs(gpu_mu0 - mu[:, 0]).max()),
                s2_rel_err=float((np.abs(gpu_s20 - s2[:, 0]) / np.maximum(np.abs(s2[:, 0]), 1e-12)).max()))


# ---------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without torchrun


def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start ``n`` ranks of this script as fresh child processes (one per GPU), wait for them, print rank 0's JSON
    line.  Runs BEFORE anything in this process has touched the GPU (no torch import here).  Returns the exit code."""
    import subprocess

    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:  # exactly the processes started above
                    procs[q].terminate()
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    lines = [ln for ln in out0 if ln.strip().startswith("{")]
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------


def _device_identity(torch, local_rank):
    """Something that tells two GPUs apart: the device's UUID when the runtime reports one, else its PCI location."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        for attr in ("uuid", "pci_bus_id"):
            v = getattr(p, attr, None)
            if v is not None:
                extra = "".join(":%s" % getattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(p, a))
                return f"{v}{extra}"
    except Exception:  # noqa: BLE001 - identity is informational
        pass
    return f"cuda:{local_rank}"


def _committed_traffic():
    """The committed PMC traffic figures (PMC cannot be collected inside a timed run): the measurement taken on THIS
    code state (tools/source_hash.py over the library's sources) if there is one, else the newest one, flagged as such."""
    import glob

    from tools.source_hash import source_hash

    cur = source_hash()
    cands = []
    for tfile in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        with open(tfile) as fh:
            cands.append(json.load(fh))
    match = [t for t in cands if t.get("source_sha256") == cur]
    if match:
        return match[-1], "measured on this code state (source sha256 %s)" % cur[:12]
    if cands:
        old = cands[-1].get("source_sha256")
        return cands[-1], "STALE: measured on %s, this run is source sha256 %s" % (
            ("source sha256 " + old[:12]) if old else "an unrecorded code state (round <= 2)", cur[:12])
    return {}, None


def both_splits(args, world, dist, dev, S_arg, evaluate, sync, grad):
    """Multi-rank runs carry BOTH splits of the batch and the single-GPU yardstick in one job, so
    that a scaling record separates "exchange cost" from "small-batch inefficiency" without a second launch.  After the
    timed region, every rank runs `steps` extra steps of each:
      other split   the split the headline did NOT use -- weak (the configuration's S samples on EVERY rank, sharded call)
                    when the headline is BASELINE's strong split, and vice versa -- max over ranks;
      expected      the same two batches of THIS rank evaluated rank-locally (gp.shard = False: no process group inside
                    the call) -- what one GPU needs for its block when nothing is exchanged -- max over ranks.
    `evaluate(hyp, shard)` runs one step; reference loops being split: f_min_fill.py:174-176, gaussian_process.py:876-879."""
    import torch

    from gpyreg_amd import sharding as _sh

    rank = dist.get_rank() if dist is not None else 0
    S_strong, S_weak = S_arg, S_arg * world
    out = {}

    def timed(hyp, shard, collective):
        evaluate(hyp, shard)  # warm-up of this shape
        if collective:
            sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            evaluate(hyp, shard)
        if collective:
            sync()
        dt = (time.perf_counter() - t0) / args.steps
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0].item())
        return dt

    other = "weak" if args.scaling != "weak" else "strong"
    S_other = S_weak if other == "weak" else S_strong
    _, _, hyp_o = synthetic_problem(args.config, S_other)
    dt = timed(hyp_o, True, True)
    per = [_sh.shard_bounds(S_other, r, world)[1] - _sh.shard_bounds(S_other, r, world)[0] for r in range(world)]
    out[other] = {"value": S_other / dt, "unit": "fit-evals/s" if grad else "NLL evals/s", "ms_per_step": dt * 1e3,
                  "global_samples": S_other, "samples_per_gpu": per[0] if len(set(per)) == 1 else per,
                  "steps": args.steps, "timing": "max over ranks, barrier + device sync on both sides"}
    exp = {}
    for name, S in (("strong", S_strong), ("weak", S_weak)):
        _, _, hyp_s = synthetic_problem(args.config, S)
        lo, hi = _sh.shard_bounds(S, rank, world)
        dt = timed(hyp_s[lo:hi], False, False) if hi > lo else 0.0
        if hi <= lo and dist is not None:  # a rank without rows still joins the reduction
            t = torch.tensor([0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0].item())
        exp[name] = {"ms_per_step": dt * 1e3, "samples": max(per_r(S, world)), "value_if_no_exchange": S / dt if dt > 0 else None}
    exp["how"] = ("this rank's block of each split evaluated rank-locally in the same process (gp.shard = False), "
                  "max over ranks; the sharded figures above minus these are the exchange and the wait for the slowest rank")
    out["expected_from_1gpu"] = exp
    return out


def per_r(S, world):
    from gpyreg_amd import sharding as _sh

    return [_sh.shard_bounds(S, r, world)[1] - _sh.shard_bounds(S, r, world)[0] for r in range(world)]


def dry_run(args, world, rank):
    """Rehearsal of the launcher and of the sharded exchange WITHOUT device work (runs on a machine without a GPU:
    tests/test_bench_launch.py): the ranks form the process group, partition the configuration's batch exactly as
    the timed path does and push stand-in rows through `sharding.gather_rows`.  No throughput is reported."""
    import torch.distributed as dist

    from gpyreg_amd import sharding as _sh

    cfg = CONFIGS[args.config]
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank):  # test hook: this rank dies before it joins the group
        sys.exit(3)
    if "RANK" in os.environ and world > 1:
        dist.init_process_group("gloo")
        world = dist.get_world_size()
    S_arg = args.samples or cfg["S"]
    S_global = S_arg * world if args.scaling == "weak" else S_arg
    _, _, hyp = synthetic_problem(args.config, S_global)
    C = 1 + hyp.shape[1]

    def local(lo, hi):  # stand-in for the device: every row is a function of its hyperparameter vector only
        rows = np.concatenate([hyp[lo:hi].sum(1, keepdims=True), 2.0 * hyp[lo:hi]], axis=1)
        return rows, np.zeros(hi - lo, bool)

    _sh.reset_stats()
    for _ in range(args.warmup + args.steps):
        full, bad = _sh.gather_rows(S_global, C, local, None, _sh.fingerprint(hyp))
    assert full.shape == (S_global, C) and np.array_equal(full[:, 1:], 2.0 * hyp) and not bad.any()
    per_rank = [_sh.shard_bounds(S_global, r, world)[1] - _sh.shard_bounds(S_global, r, world)[0] for r in range(world)]
    splits = {}
    calls_per_step = _sh.stats()["calls"] / (args.warmup + args.steps)
    if dist.is_initialized():
        import torch

        def evaluate(h, shard):  # the same stand-in rows, sharded through the exchange or rank-local
            def loc(lo, hi):
                return np.concatenate([h[lo:hi].sum(1, keepdims=True), 2.0 * h[lo:hi]], axis=1), np.zeros(hi - lo, bool)
            if shard:
                return _sh.gather_rows(h.shape[0], 1 + h.shape[1], loc, None, _sh.fingerprint(h))
            return loc(0, h.shape[0])

        splits = both_splits(args, world, dist, torch.device("cpu"), S_arg, evaluate, dist.barrier, True)
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        extra = {k: v for k, v in splits.items()}
        print(json.dumps({
            "metric": "launcher rehearsal (no device work)", "value": None, "unit": None, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "scaling": "weak" if args.scaling == "weak" else "strong",
            "data": "dry run", "config": {
                "workload": f"cfg{args.config} batch partition only", "global_samples": S_global,
                "samples_per_gpu": per_rank[0] if len(set(per_rank)) == 1 else per_rank,
                "launcher": os.environ.get("BENCH_LAUNCHER", "env" if "RANK" in os.environ else "single process"),
                "backend": "gloo" if dist.is_initialized() else None, "process_group_ranks": world,
                "exchanges_per_step": calls_per_step, **extra}}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = CONFIGS[args.config]
    dtype = args.dtype or ("f32" if args.config == 4 else "f64")
    mode = "nll" if args.nll_only else args.mode

    import torch

    if args.dry_run:
        return dry_run(args, world, rank)
    dist = None
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank = local_rank % max(ndev, 1)  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    if "RANK" in os.environ and (world > 1 or os.environ.get("BENCH_FORCE_DIST")):
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()  # what the process group says, not what the command line asked for
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    # the batch: BASELINE.json's own split by default ("config": the configuration's S samples over the ranks), or
    # the configuration's S samples on EVERY rank ("weak")
    S_arg = args.samples or cfg["S"]
    S_global = S_arg * world if args.scaling == "weak" else S_arg
    from gpyreg_amd import _lib
    from gpyreg_amd import sharding as _sh

    lo, hi = _sh.shard_bounds(S_global, rank, world)
    S_local = hi - lo
    per_rank = [_sh.shard_bounds(S_global, r, world)[1] - _sh.shard_bounds(S_global, r, world)[0] for r in range(world)]

    devices = [_device_identity(torch, local_rank)]
    if dist is not None:
        ids = [None] * world
        dist.all_gather_object(ids, (os.uname().nodename, devices[0]))
        devices = ["%s/%s" % t for t in ids]
        if args.backend == "nccl":
            assert len(set(devices)) == world, f"{world} ranks but the devices are {devices}"

    X, y, hyp = synthetic_problem(args.config, S_global)  # the global batch, identical on every rank
    gp = make_gp(args.config, dtype)
    gp.device = local_rank
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)  # X, y -> HBM on first use
    ctx = _lib.context(local_rank)
    grad = mode == "fit"
    xs = None
    if mode == "predict":
        # the posteriors of the S samples (sharded like the batch: each rank factors and keeps its block), untimed
        gp.update(hyp=hyp)
        xs = np.random.default_rng(77).uniform(-3, 3, (args.points, cfg["D"]))

    def step():
        # with a process group: sharded over the ranks and all-gathered inside the GP's own methods
        if mode == "predict":
            return gp.predict(xs, separate_samples=True)
        return gp.nll_batch(hyp, compute_grad=grad)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    _sh.reset_stats()
    fac_ms, tot_ms, lau_in = [], [], []
    lau_fl = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r0, r1 = step()
        a, b = ctx.last_timing()  # hipEvents on the library's own stream
        tot_ms.append(a)
        fac_ms.append(b)
        if grad:
            lm, lf = ctx.last_lauum_timing()  # the W^T W launch of this step, as it ran inside the pipeline
            if lf > 0:
                lau_in.append(lm)
                lau_fl = lf
    sync()
    dt = time.perf_counter() - t0
    exch = _sh.stats()
    if dist is not None:
        tmax = torch.tensor([dt, exch["seconds"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        exch_max = float(tmax[1].item())
        assert r0.shape[-1 if mode == "predict" else 0] == S_global  # every rank holds every sample's result
    else:
        exch_max = exch["seconds"]

    # both splits and the single-GPU yardstick in the same job (multi-rank runs; after the timed region)
    splits = {}
    if dist is not None and world > 1 and mode != "predict":
        def evaluate(h, shard):
            prev = gp.shard
            gp.shard = shard
            try:
                return gp.nll_batch(h, compute_grad=grad)
            finally:
                gp.shard = prev

        splits = both_splits(args, world, dist, dev, S_arg, evaluate, sync, grad)

    # The dominant single kernel, timed ALONE as well: three extra UNTIMED steps with one sample group and the two
    # triangular mat-vecs after the launch instead of under it (they cost it ~3 %), so that its hipEvent
    # time is the kernel's own duration; the first of the three is dropped.
    lau_alone, lau_alone_fl = [], 0.0
    if rank == 0 and grad and S_local > 0:
        prev = {k: ctx.get_option(k) for k in ("groups", "solves_beside_lauum")}
        ctx.set_option("groups", 1)
        ctx.set_option("solves_beside_lauum", 0)
        gp.shard = False  # rank-local extra steps (the other ranks are not in this loop)
        for _ in range(3):
            gp.nll_batch(hyp[lo:hi], compute_grad=True)
            lm, lf = ctx.last_lauum_timing()
            if lf > 0:
                lau_alone.append(lm)
                # the flops of THIS launch: with one sample group forced, a batch that the timed loop runs as two groups
                # (cfg5: 2 x 32 samples) is one launch of all its samples here (dividing the timed loop's
                # per-group flops by this launch's time read 0.45 where the launch ran at 0.90)
                lau_alone_fl = lf
        lau_alone = lau_alone[1:]
        for k, v in prev.items():
            ctx.set_option(k, v)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    N, D = cfg["N"], cfg["D"]
    peak = FP64_MFMA_PEAK_TFLOPS if dtype == "f64" else FP32_MFMA_PEAK_TFLOPS
    tf, cyc, ghz = ctx.mfma_peak(_lib.F64 if dtype == "f64" else _lib.F32)
    wall = dt / args.steps
    fac = float(np.mean(fac_ms)) * 1e-3
    workload = (f"cfg{args.config}: N={N} D={D} {cfg['kernel']}{cfg['degree'] or ''} ARD, ConstantMean, "
                f"GaussianNoise(constant), ")
    out = {
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if args.scaling == "weak" else "strong",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
    }
    config = {
        "global_samples": S_global,
        "samples_per_gpu": per_rank[0] if len(set(per_rank)) == 1 else per_rank,
        "sharding": f"hyperparameter samples over {world} rank(s), block partition"
                    + ("; BASELINE.json's split (total work fixed)" if args.scaling == "config" else "; S per GPU fixed"),
        "launcher": os.environ.get("BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else
                                   ("env" if "RANK" in os.environ else "single process")),
        "backend": None if dist is None else args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
        "process_group_ranks": world,
        "devices": devices,
        "distinct_devices": len(set(devices)),
        # the exchange step of a sharded call, as rank 0 / the slowest rank saw it: waiting for the slowest rank,
        # the all-gather of the frame (header + rows, gpyreg_amd/sharding.py) and its host <-> device copies
        "exchange_ms_per_step": None if dist is None else exch["seconds"] / args.steps * 1e3,
        "exchange_ms_per_step_max_over_ranks": None if dist is None else exch_max / args.steps * 1e3,
        "exchanges_per_step": None if dist is None else exch["calls"] / args.steps,
    }
    config.update(splits)  # "weak" (or "strong") and "expected_from_1gpu" when there is more than one rank

    if mode == "predict":
        mu, s2 = r0, r1
        M = args.points
        # F = N^2 M (the triangular product V = W Ks) + 2 N M (D + 1) per posterior sample
        flops_gemm = float(S_local) * float(N) * N * M
        flops_all = flops_gemm + float(S_local) * 2.0 * N * M * (D + 1)
        out.update({
            "metric": f"GP predict point-samples/sec (N={N} D={D}, M={M} query points)",
            "value": float(M) * S_global / wall,
            "unit": "point-samples/s",
        })
        config["workload"] = workload + f"GP.predict(x_star[{M}], separate_samples=True) for {S_global} posterior samples"
        config["query_points"] = M
        out["config"] = config
        out["roofline"] = {
            "bound": "mfma",
            "kernel": "V = W Ks: gemm_kernel / gemm_persist_kernel<T,false,true,...> with k <= row tile (gaussian_process.py:1752-1760 as a product)",
            "achieved": flops_gemm / fac / 1e12 if fac > 0 else None,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": flops_gemm / fac / 1e12 / peak if fac > 0 else None,
            "traffic": None,
            "flops_per_launch": flops_gemm,
            "launch_ms": fac * 1e3,
            "device_ms_per_step": float(np.mean(tot_ms)),
            # the whole call by the wall clock (cross-kernel build, mat-vec, product, column sums, transfers, host)
            "frac_wall": flops_all / wall / 1e12 / peak,
            "hbm_bytes_algorithmic": float(S_local) * N * N * (8 if dtype == "f64" else 4) / 2,
            "measured_mfma_ceiling": {"tflops": tf, "cycles_per_mfma_per_simd": cyc, "clock_ghz": ghz},
        }
        out["mu_sample0_first"] = float(mu[0, 0])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_predict(args.config, xs, mu[:, 0], s2[:, 0])
            out["vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if dtype == "f64":
                assert out["cpu_baseline"]["mu_abs_err"] < 1e-8 and out["cpu_baseline"]["s2_rel_err"] < 1e-6, out["cpu_baseline"]
        print(json.dumps(out), flush=True)
    else:
        nlz, dnlz = r0, r1
        fits = S_global * args.steps
        flops_step = S_local * (float(N) ** 3 if grad else float(N) ** 3 / 3.0)  # this rank's share
        traffic, traffic_state = _committed_traffic()
        quoted = args.config == 3 and grad and dtype == "f64" and world == 1 and S_global == cfg["S"]
        out.update({
            "metric": "GP-fits/sec (NLL+grad, N=4096 D=10)" if (args.config == 3 and grad)
            else f"GP {'fits' if grad else 'NLL evals'}/sec (N={N} D={D})",
            "value": fits / dt,
            "unit": "fit-evals/s",
        })
        config["workload"] = workload + f"NLL{'+grad' if grad else ''}"
        out["config"] = config
        have_dom = grad and lau_fl > 0 and lau_in
        roof = {
            "bound": "mfma",
            "unit": "TFLOP/s",
            "peak": peak,
            "measured_mfma_ceiling": {"tflops": tf, "cycles_per_mfma_per_simd": cyc, "clock_ghz": ghz},
            # the whole pipeline of one step on this rank: S N^3 algorithmic flops
            "flops_per_step": flops_step,
            "factor_section_ms": fac * 1e3,
            "device_ms_per_step": float(np.mean(tot_ms)),
            # ... over the factorization section (hipEvents around the gemm / leaf launch sequence of plan.h) ...
            "frac_factor_section": flops_step / fac / 1e12 / peak if fac > 0 else None,
            # ... and over the wall clock of the timed region (the one number the driver also times)
            "frac_wall": flops_step / wall / 1e12 / peak,
            # memory-side bytes per step from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled,
            # + WRITE_SIZE; profiles/r*_pmc_summary.txt)
            "traffic_per_step": traffic.get("step_traffic_bytes") if quoted else None,
            "traffic_source": traffic.get("source") if quoted else None,
            "traffic_code_state": traffic_state if quoted else None,
        }
        if have_dom:
            # the dominant kernel: W^T W ("lauum"), gemm_persist_kernel<T,true,true,128,4> in the rocprofv3 summary
            # under profiles/; one launch per step with all samples of the rank
            lm = float(np.mean(lau_in)) * 1e-3
            roof.update({
                "kernel": "gemm_persist_kernel<T,true,true,128,4> (lauum: (K+sn2 I)^-1 = W^T W), all samples in one persistent launch",
                "flops_per_launch": lau_fl,
                "launch_ms": lm * 1e3,
                "launch_ms_schedule": "as it runs in the timed steps (the two triangular mat-vecs co-resident under it), "
                                      "hipEvents on its stream, mean of the timed steps",
                "achieved": lau_fl / lm / 1e12,
                "frac": lau_fl / lm / 1e12 / peak,
                "traffic": traffic.get("dominant_kernel_traffic_bytes") if quoted else None,
            })
            if lau_alone:
                la = float(np.mean(lau_alone)) * 1e-3
                roof["alone"] = {"launch_ms": la * 1e3, "flops_per_launch": lau_alone_fl,
                                 "achieved": lau_alone_fl / la / 1e12, "frac": lau_alone_fl / la / 1e12 / peak,
                                 "launch_ms_schedule": "untimed extra steps with one sample group and the mat-vecs AFTER "
                                                       "the launch (overlap off)"}
        else:
            roof.update({
                "kernel": "blocked potrf" + (" + trtri + lauum" if grad else "") + " (gemm_kernel + leaf_kernel launches of one batch)",
                "flops_per_launch": flops_step,
                "launch_ms": fac * 1e3,
                "achieved": flops_step / fac / 1e12 if fac > 0 else None,
                "frac": flops_step / fac / 1e12 / peak if fac > 0 else None,
                "traffic": None,
            })
        out["roofline"] = roof
        out["nlz_sample0"] = float(nlz[0])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, float(nlz[0]), dnlz[0] if grad else None, grad)
            out["vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if grad and args.config in (2, 3) and dtype == "f64":  # north_star bar, checked live
                assert out["cpu_baseline"]["nlz_rel_err"] < 1e-8 and out["cpu_baseline"]["grad_rel_err"] < 1e-8, \
                    out["cpu_baseline"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="fit", choices=["fit", "nll", "predict"])
    ap.add_argument("--scaling", default="config", choices=["config", "weak"],
                    help="config: the configuration's S samples split over the GPUs (BASELINE.json; total work fixed); "
                         "weak: S samples on every GPU")
    ap.add_argument("--samples", type=int, default=0,
                    help="hyperparameter samples of the batch (default: the configuration's S): global with --scaling "
                         "config, per GPU with --scaling weak")
    ap.add_argument("--points", type=int, default=1000, help="--mode predict: query points M")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nll-only", action="store_true", help="same as --mode nll")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / exchange rehearsal without device work (no GPU needed, gloo); reports no throughput")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us: start the ranks ourselves, before this process touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    run(args)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- GP fit-evaluations/sec (NLL + gradient) on MI355X.

A "step" is one batched pass of the hot path: `GP.nll_batch(hyp[S], compute_grad=True)`
for S hyperparameter samples at the workload BASELINE.json's metric is quoted on
(N=4096, D=10, Matern-5/2 ARD, ConstantMean, constant Gaussian noise, fp64).
X and y are resident in HBM before the timed region; the per-step H2D of the
hyperparameter-derived vectors and D2H of (nlZ, dnlZ) are inside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU; samples are independent units.  Every rank passes the SAME
global batch of world x S hyperparameter vectors to `GP.nll_batch`, which block-partitions
them over the process group (gpyreg_amd/sharding.py: S per rank, "weak" scaling), evaluates its
block on its own GPU and all-gathers the per-sample [nlZ | dnlZ] rows over RCCL -- the one
exchange step of the path, done every step, inside the product's own API.  Rank 0 prints
ONE JSON line.
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


def _oracle_eval_seconds(cfg_idx, repeats, warmup, grad=True):
    """Wall-clock seconds of ``repeats`` evaluations of sample 0 by the CPU oracle (after ``warmup``
    untimed ones), plus the last result.  Also the body of the 1-thread child process."""
    from oracle import gp_oracle as orc  # cpu_baseline leg only

    c = CONFIGS[cfg_idx]
    X, y, hyp = synthetic_problem(cfg_idx, 1)
    model = dict(kernel=c["kernel"], degree=c["degree"], mean="const", noise=(1, 0, 0))
    times, out = [], None
    for it in range(warmup + repeats):
        t0 = time.perf_counter()
        out = orc.core(model, hyp[0], X, y, None, 1, 1 if grad else 0)
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    return times, (out if grad else (out, None))


def cpu_baseline(cfg_idx, gpu_nlz0=None, gpu_dnlz0=None):
    """The CPU oracle (NumPy/SciPy restatement pinned to the reference) timed on this host's
    cores on a BOUNDED sample of the same workload: sample 0 of the batch, NLL+gradient, one
    warm-up evaluation then the median of three (default BLAS threading = all cores), and one
    evaluation in a child process restricted to ONE BLAS thread.  The oracle's value and gradient
    double as a live parity check of the GPU result (``grad_rel_err``)."""
    import subprocess

    c = CONFIGS[cfg_idx]
    # cfg2/cfg3: the full protocol.  cfg5 (N=8192): one NLL+grad evaluation (minutes).  cfg4
    # (N=16384, RQ): the reference has no fp32 path and its (N,N,22) gradient tensor needs 47 GB,
    # so the CPU figure is one fp64 NLL-only evaluation.
    full = cfg_idx in (2, 3)
    grad = cfg_idx != 4
    times, (nlz, dnlz) = _oracle_eval_seconds(cfg_idx, repeats=3 if full else 1, warmup=1 if full else 0, grad=grad)
    med = float(np.median(times))
    if not grad:
        gpu_dnlz0 = None
    one = None
    try:
        if not full:
            raise RuntimeError("single-thread figure only for cfg2/cfg3")
        env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "t, _ = bench._oracle_eval_seconds(%d, 1, 0); print(json.dumps(t))" % (ROOT, cfg_idx))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        one = float(json.loads(r.stdout.strip().splitlines()[-1])[0])
    except Exception:  # noqa: BLE001 - the 1-thread figure is optional
        one = None
    host = _host_description()
    # (the BLAS pools only: an OpenMP runtime that torch brought into the process takes no part in the oracle's work)
    blas_threads = max([b.get("num_threads") or 1 for b in host["blas"]
                        if b.get("internal_api") in ("openblas", "mkl", "blis", "flexiblas")] or [1])
    # `cores` = the threads the evaluation can actually use: the BLAS pool (NumPy's elementwise passes, which
    # dominate, are single-threaded); the logical CPU count is in host.logical_cpus
    out = dict(value=1.0 / med, unit="fit-evals/s", cores=blas_threads, kind="port",
               sample=f"sample 0 of {c['S']}, {'NLL+grad' if grad else 'NLL only (fp64)'}, N={c['N']} D={c['D']} "
                      f"{c['kernel']}{c['degree'] or ''}: {'1 warm-up + median of 3 evaluations' if full else '1 evaluation'} "
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--samples", type=int, default=0, help="hyperparameter samples per GPU (default: config's S)")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nll-only", action="store_true", help="time NLL without gradient")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = CONFIGS[args.config]
    S = args.samples or cfg["S"]
    dtype = args.dtype or ("f32" if args.config == 4 else "f64")

    import torch

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
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    X, y, hyp = synthetic_problem(args.config, S * world)  # the global batch, identical on every rank
    gp = make_gp(args.config, dtype)
    gp.device = local_rank
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)  # X, y -> HBM on first use
    from gpyreg_amd import _lib

    ctx = _lib.context(local_rank)
    grad = not args.nll_only
    hyp_N = hyp.shape[1]
    gathered = None

    def step():
        # with a process group: sharded over the ranks and all-gathered inside GP.nll_batch
        nonlocal gathered
        nlz, dnlz = gp.nll_batch(hyp, compute_grad=grad)
        gathered = nlz
        return nlz, dnlz

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    fac_ms, tot_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nlz, dnlz = step()
        a, b = ctx.last_timing()  # hipEvents on the library's own stream
        tot_ms.append(a)
        fac_ms.append(b)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        assert gathered.shape == (world * S,)  # every rank holds every sample's result

    # The dominant single kernel, timed alone: two extra UNTIMED steps with one sample group, so
    # the W^T W launch (gemm_persist_kernel<T,true,true,128,4>, all S samples in one grid) is not
    # co-scheduled with another group's kernels and its hipEvent time is the kernel's duration.
    lau_ms, lau_fl = [], 0.0
    if rank == 0 and grad:
        groups_env = int(os.environ.get("GPC_GROUPS", "2"))
        ctx.set_option("groups", 1)
        # ... nor with the two triangular mat-vecs that the timed steps run UNDER it (they cost it ~3 %)
        ctx.set_option("solves_beside_lauum", 0)
        gp.shard = False  # rank-local extra steps (the other ranks are not in this loop)
        for _ in range(3):
            gp.nll_batch(hyp[:S], compute_grad=True)
            lm, lau_fl = ctx.last_lauum_timing()
            lau_ms.append(lm)
        lau_ms = lau_ms[1:]
        ctx.set_option("groups", groups_env)
        ctx.set_option("solves_beside_lauum", 1)

    if rank == 0:
        N = cfg["N"]
        fits = S * args.steps * world
        flops_per_launch = S * (float(N) ** 3 if grad else float(N) ** 3 / 3.0)
        fac = float(np.mean(fac_ms)) * 1e-3
        peak = FP64_MFMA_PEAK_TFLOPS if dtype == "f64" else FP32_MFMA_PEAK_TFLOPS
        achieved = flops_per_launch / fac / 1e12
        from gpyreg_amd import _lib as L_

        tf, cyc, ghz = ctx.mfma_peak(L_.F64 if dtype == "f64" else L_.F32)
        # Committed PMC traffic (PMC cannot be collected inside a timed run): the measurement taken on THIS code state
        # (tools/source_hash.py over the library's sources) if there is one, else the newest one, flagged as such
        traffic, traffic_state = {}, None
        import glob
        from tools.source_hash import source_hash

        cur = source_hash()
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        cands = []
        for tfile in tfiles:
            with open(tfile) as fh:
                cands.append(json.load(fh))
        match = [t for t in cands if t.get("source_sha256") == cur]
        if match:
            traffic, traffic_state = match[-1], "measured on this code state (source sha256 %s)" % cur[:12]
        elif cands:
            traffic = cands[-1]
            old = traffic.get("source_sha256")
            traffic_state = "STALE: measured on %s, this run is source sha256 %s" % (
                ("source sha256 " + old[:12]) if old else "an unrecorded code state (round <= 2)", cur[:12])
        out = {
            "metric": "GP-fits/sec (NLL+grad, N=4096 D=10)" if (args.config == 3 and grad)
            else f"GP {'fits' if grad else 'NLL evals'}/sec (N={N} D={cfg['D']})",
            "value": fits / dt,
            "unit": "fit-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {
                "workload": f"cfg{args.config}: N={N} D={cfg['D']} {cfg['kernel']}{cfg['degree'] or ''} ARD, "
                            f"ConstantMean, GaussianNoise(constant), NLL{'+grad' if grad else ''}",
                "samples_per_gpu": S,
                "global_samples": S * world,
                "sharding": f"hyperparameter samples x{world}",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "blocked potrf + trtri + lauum (gemm_kernel + leaf_kernel launches of one batch)",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                # memory-side bytes per step from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled,
                # + WRITE_SIZE; profiles/r*_pmc_summary.txt); PMC cannot be collected inside a timed run
                "traffic": traffic.get("step_traffic_bytes") if (args.config == 3 and grad and dtype == "f64") else None,
                "traffic_source": traffic.get("source") if (args.config == 3 and grad and dtype == "f64") else None,
                "traffic_code_state": traffic_state if (args.config == 3 and grad and dtype == "f64") else None,
                "flops_per_launch": flops_per_launch,
                "launch_ms": fac * 1e3,
                "device_ms_per_step": float(np.mean(tot_ms)),
                "measured_mfma_ceiling": {"tflops": tf, "cycles_per_mfma_per_simd": cyc, "clock_ghz": ghz},
                # the single dominant kernel (one launch per sample group): W^T W ("lauum"),
                # gemm_persist_kernel<T,true,true,128,4> in the rocprofv3 summary under profiles/
                "dominant_kernel": None if not (grad and lau_fl > 0) else {
                    "name": "gemm_persist_kernel<T,true,true,128,4> (lauum: (K+sn2 I)^-1 = W^T W), all samples in one persistent launch",
                    "flops_per_launch": lau_fl,
                    "launch_ms": float(np.mean(lau_ms)),
                    "achieved": lau_fl / (float(np.mean(lau_ms)) * 1e-3) / 1e12,
                    "frac": lau_fl / (float(np.mean(lau_ms)) * 1e-3) / 1e12 / peak,
                    "traffic": traffic.get("dominant_kernel_traffic_bytes") if (args.config == 3 and dtype == "f64") else None,
                },
            },
            "nlz_sample0": float(nlz[0]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, float(nlz[0]), dnlz[0] if grad else None)
            out["vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if grad and args.config in (2, 3) and dtype == "f64":  # north_star bar, checked live
                assert out["cpu_baseline"]["nlz_rel_err"] < 1e-8 and out["cpu_baseline"]["grad_rel_err"] < 1e-8, \
                    out["cpu_baseline"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""`bench.py --gpus N` starts its own ranks (no torchrun) and splits the batch as BASELINE.json does.

CPU part: the launcher, the partition and the sharded exchange through `--dry-run` (gloo, no device work).
GPU part: two self-launched ranks sharing the one GPU of the box run the real timed path."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                       timeout=timeout, env=e, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r, line = _bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["config"]["process_group_ranks"] == 2
    assert line["config"]["launcher"] == "self"
    # BASELINE.json cfg3: 16 samples in all, split over the GPUs -- total work fixed
    assert line["config"]["global_samples"] == 16 and line["config"]["samples_per_gpu"] == 8
    assert line["scaling"] == "strong"
    assert len(r.stdout.strip().splitlines()) == 1  # ONE line on stdout
    assert line["config"]["exchanges_per_step"] == 1.0  # one collective per sharded call (the frame carries the rows)
    _check_both_splits(line, world=2, S=16, headline="strong")


def _check_both_splits(line, world, S, headline):
    """The record of a multi-rank run carries the OTHER split and the single-GPU yardstick."""
    cfg = line["config"]
    other = "weak" if headline == "strong" else "strong"
    assert other in cfg and headline not in cfg
    o = cfg[other]
    S_other = S * world if other == "weak" else S
    assert o["global_samples"] == S_other and o["value"] > 0 and o["ms_per_step"] > 0
    assert abs(o["value"] - S_other / (o["ms_per_step"] * 1e-3)) < 1e-6 * o["value"]
    e = cfg["expected_from_1gpu"]
    assert e["strong"]["samples"] == -(-S // world) and e["weak"]["samples"] == S
    for k in ("strong", "weak"):
        assert e[k]["ms_per_step"] > 0 and e[k]["value_if_no_exchange"] > 0


def test_config_split_at_eight_ranks_and_weak_scaling():
    r, line = _bench("--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 8 and line["config"]["global_samples"] == 16 and line["config"]["samples_per_gpu"] == 2
    r, line = _bench("--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "0", "--config", "5")
    assert line["config"]["global_samples"] == 64 and line["config"]["samples_per_gpu"] == 8
    r, line = _bench("--gpus", "4", "--dry-run", "--steps", "1", "--warmup", "0", "--scaling", "weak")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 4 and line["config"]["global_samples"] == 64 and line["config"]["samples_per_gpu"] == 16
    assert line["scaling"] == "weak"
    _check_both_splits(line, world=4, S=16, headline="weak")


def test_a_failing_rank_fails_the_launch():
    r, line = _bench("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", env={"BENCH_TEST_FAIL_RANK": "1"},
                     timeout=120)
    assert r.returncode != 0 and line is None
    assert "rank 1 exited with code 3" in r.stderr


def test_under_a_launcher_the_environment_is_used():
    # what torchrun does: RANK / WORLD_SIZE set from outside, one process per rank
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items()}
    procs = []
    for rk in range(2):
        env = dict(base, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                                       "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    line = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["launcher"] == "env"
    assert not [ln for ln in outs[1].splitlines() if ln.startswith("{")]  # only rank 0 reports


@pytest.mark.gpu
def test_two_self_launched_ranks_on_one_gpu_run_the_timed_path():
    r, line = _bench("--gpus", "2", "--backend", "gloo", "--config", "2", "--samples", "4", "--steps", "2", "--warmup", "1",
                     "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["config"]["samples_per_gpu"] == 2 and line["config"]["global_samples"] == 4
    assert line["value"] > 0 and line["config"]["exchanges_per_step"] == 1.0
    assert line["config"]["launcher"] == "self" and line["scaling"] == "strong"
    _check_both_splits(line, world=2, S=4, headline="strong")
    r1, one = _bench("--gpus", "1", "--config", "2", "--samples", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r1.returncode == 0, r1.stderr[-3000:]
    assert one["n_gpus"] == 1 and abs(one["nlz_sample0"] - line["nlz_sample0"]) == 0.0  # same bits sharded or not


@pytest.mark.gpu
def test_predict_mode_reports_a_roofline_and_matches_the_oracle():
    r, line = _bench("--mode", "predict", "--config", "2", "--samples", "3", "--points", "300", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["unit"] == "point-samples/s" and line["value"] > 0
    assert line["roofline"]["launch_ms"] > 0 and 0 < line["roofline"]["frac"] < 1
    assert line["cpu_baseline"]["mu_abs_err"] < 1e-8 and line["cpu_baseline"]["s2_rel_err"] < 1e-6


@pytest.mark.gpu
def test_the_rccl_branch_with_one_rank():
    """One GPU cannot hold two nccl ranks, but the code the 8-GPU run goes through -- `init_process_group("nccl")`, device
    tensors in `sharding._Gather`, `all_gather_into_tensor` / `all_gather_object` / `all_reduce` on the device, the
    barrier before the group is destroyed -- runs with a group of ONE rank too: the bench under BENCH_FORCE_DIST, and the
    gather on device tensors directly."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
           "BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r, line = _bench("--gpus", "1", "--config", "2", "--samples", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                     env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 1 and line["config"]["backend"].startswith("nccl") and line["value"] > 0
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "from gpyreg_amd import sharding as sh\n"
        "a = np.arange(12.0).reshape(4, 3)\n"
        "g = sh._Gather(a, 4)\n"
        "assert g.bufs[1].is_cuda and np.array_equal(g.result(), a)\n"
        "dist.barrier(); dist.destroy_process_group(); print('gather ok')\n" % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, **dict(env, MASTER_PORT=str(port)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert p.returncode == 0 and "gather ok" in p.stdout, p.stderr[-3000:]


@pytest.mark.gpu
def test_direct_rccl_exchange_one_rank():
    """GPYREG_AMD_EXCHANGE=rccl (optional, off by default): the exchange as ONE direct `ncclAllGather` through ctypes on a
    communicator made from the process group (gpyreg_amd/_rccl.py) -- the same rows as the torch.distributed exchange, buffers
    reused across many gathers of several shapes, and the frame of `gather_rows`."""
    import socket

    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "from gpyreg_amd import sharding as sh, _rccl\n"
        "rng = np.random.default_rng(3)\n"
        "for k in range(40):\n"
        "    rows, cols = [(1, 512), (4, 3), (2, 15), (16, 2001)][k %% 4]\n"
        "    a = rng.standard_normal((rows, cols))\n"
        "    g = sh._Gather(a, rows)\n"
        "    assert g.direct is not None and np.array_equal(g.result(), a), k\n"
        "assert not _rccl._state['failed'] and len(_rccl._state['comms']) == 1\n"
        "dist.barrier(); dist.destroy_process_group(); print('direct gather ok')\n" % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
             HSA_ENABLE_IPC_MODE_LEGACY="0", GPYREG_AMD_EXCHANGE="rccl")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert p.returncode == 0 and "direct gather ok" in p.stdout, p.stderr[-3000:]


def test_under_torch_distributed_run_exactly_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...` (the multi-GPU command of the bench contract), here with --dry-run: no second launch of ranks, the
    launcher's environment is used, rank 0 prints the one line."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["launcher"] in ("torchrun", "env")
    assert line["config"]["global_samples"] == 16 and line["config"]["samples_per_gpu"] == 8


@pytest.mark.gpu
def test_the_bench_line_carries_every_field_of_the_contract():
    """One line of JSON with the fields the driver and the judge read (metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, `roofline` with bound,
    achieved, peak, unit, frac and traffic, `cpu_baseline` with value, unit, cores, kind and sample) -- on cfg2, whose CPU
    leg takes seconds."""
    r, line = _bench("--config", "2", "--steps", "5", "--warmup", "2")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1  # ONE line on stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_factor_section", "frac_wall"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["nlz_rel_err"] < 1e-8 and cb["grad_rel_err"] < 1e-8
    assert abs(line["value"] - 1.0 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]  # one sample per step
    assert "10 evaluations" in cb["sample"] and "3 warm-up" in cb["sample"]  # BASELINE.md's protocol where it is cheap


@pytest.mark.gpu
def test_the_alone_figure_of_a_batch_that_runs_as_two_sample_groups():
    """`roofline.alone` times the dominant launch in extra steps with ONE sample group; a batch beyond S (N_pad/4096)^3 = 64
    runs its timed steps as TWO groups (two W^T W launches of S/2 samples each).  The alone figure must be priced with the
    flops of the launch it timed (all S samples), not with the timed loop's per-group flops -- round 5 read 0.45 for a
    launch that ran at 0.90.  cfg5's size, N = 8192, with 16 samples (S (N_pad/4096)^3 = 128: two groups
    of 8; beyond N_pad = 4096 the one-group pipeline is issued eagerly, so its launch carries events)."""
    r, line = _bench("--config", "5", "--samples", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    roof = line["roofline"]
    alone = roof["alone"]
    assert abs(alone["flops_per_launch"] - 2.0 * roof["flops_per_launch"]) < 1e-6 * alone["flops_per_launch"]
    assert alone["frac"] >= roof["frac"] - 0.02, (alone["frac"], roof["frac"])
    assert 0.5 < alone["frac"] < 1.0

"""CPU-side checks of the drop-in boundary: the shared library builds, loads, exports
every symbol include/gpcore.h declares, and the product path fails loudly without a GPU."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    from gpyreg_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "gpcore.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    # declarations under GPC_EXPERIMENTS belong to the experiments build (lib/libgpcore_exp.so), not to the product
    experiments = "".join(re.findall(r"#ifdef GPC_EXPERIMENTS(.*?)#endif", header, flags=re.S))
    product = re.sub(r"#ifdef GPC_EXPERIMENTS.*?#endif", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gpc_[A-Za-z0-9_]+)\s*\(", product))
    declared_exp = set(re.findall(r"\b(gpc_[A-Za-z0-9_]+)\s*\(", experiments))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert declared_exp == set(_lib.EXPERIMENT_SIGNATURES), declared_exp ^ set(_lib.EXPERIMENT_SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    if os.path.basename(_lib.LIB_PATH) == "libgpcore.so":  # the product exports none of the experiment entries
        for name in declared_exp:
            assert not hasattr(lib, name), name


def test_the_product_library_holds_the_shipped_schedule_only():
    """The schedules that were measured and rejected (dataflow graph, independent pipelines, rectangular tiles,
    right-looking panels) are compiled under GPC_EXPERIMENTS only: the product library has neither
    their kernels nor their option names."""
    import subprocess

    from gpyreg_amd import _lib

    if os.path.basename(_lib.LIB_PATH) != "libgpcore.so":
        pytest.skip("an alternative library was selected through GPYREG_AMD_LIB")
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    raw = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"dag_worker_kernel", b"dag_leaf_kernel", b"dag_small_tiles", b"indep_min_tiles", b"rect_min", b"rl_panel"):
        assert needle not in raw, needle
    assert "gpc_debug_dag" not in syms


def test_cov_count_entry_point_without_gpu():
    from gpyreg_amd import _lib

    lib = _lib.load()
    assert lib.gpc_cov_count(_lib.K_SE, 5) == 6
    assert lib.gpc_cov_count(_lib.K_RQ, 5) == 7
    assert lib.gpc_cov_count(_lib.K_MATERN_ISO, 5) == 2
    assert lib.gpc_cov_count(99, 5) == -1


def test_size_envelope_entry_point_without_gpu():
    """gpc_max_n is a memory-budget answer (round 6): the largest multiple of 128 whose three padded slabs of one sample
    fit in 80 % of the device's memory; without a device it assumes the MI355X's 288 GB.  (Rounds 1-5: 16384 / 23168,
    the reach of a 32-bit byte offset over one operand panel.)"""
    from gpyreg_amd import _lib

    lib = _lib.load()
    for dtype, w in ((_lib.F64, 8), (_lib.F32, 4)):
        n = lib.gpc_max_n(dtype)
        assert n % 128 == 0 and n > 23168
        if not _has_gpu():
            total = 288 * 2**30
            assert 3 * n * n * w <= 0.8 * total < 3 * (n + 128) * (n + 128) * w


def test_assigning_data_attributes_marks_the_device_copy_stale():
    """gp.X = ..., gp.y = ... (the reference's tests assign them directly) must invalidate the
    resident copy; invalidate() covers in-place edits."""
    import gpyreg_amd as gpr

    gp = gpr.GP(2, gpr.covariance_functions.SquaredExponential(), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp._token = object()
    gp.y = np.zeros((3, 1))
    assert gp._token is None and gp.y.shape == (3, 1)
    gp._token = object()
    gp.X = np.zeros((3, 2))
    assert gp._token is None
    gp._token = object()
    gp.s2 = np.ones((3, 1))
    assert gp._token is not None  # s2 is not resident: it reaches the device through the noise values
    gp.invalidate()
    assert gp._token is None


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_product_path_fails_loudly_without_gpu():
    import gpyreg_amd as gpr

    cov = gpr.covariance_functions.SquaredExponential()
    with pytest.raises(RuntimeError) as e:
        cov.compute(np.zeros(3), np.zeros((4, 2)))
    assert "no CPU fallback" in str(e.value)
    gp = gpr.GP(2, cov, gpr.mean_functions.ConstantMean(), gpr.noise_functions.GaussianNoise(constant_add=True))
    with pytest.raises(RuntimeError):
        gp.update(X_new=np.zeros((4, 2)), y_new=np.zeros(4), hyp=np.zeros((1, 5)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gpyreg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), os.path.join(dirpath, f)


def test_mean_and_noise_plugins_match_golden_semantics(core_golden):
    """The host-evaluated boundary plugins (mean/noise) against the oracle (itself pinned
    bit-exactly to the reference): values, gradients, scalar-vs-array return."""
    import gpyreg_amd as gpr
    from conftest import parse_core_name
    from oracle import gp_oracle as orc

    g = core_golden
    means = {"zero": gpr.mean_functions.ZeroMean, "const": gpr.mean_functions.ConstantMean,
             "negquad": gpr.mean_functions.NegativeQuadratic}
    for name in g["names"]:
        tag, model, N, D, _ = parse_core_name(name)
        X, y, hyp = g[tag + "_X"], g[tag + "_y"], g[tag + "_hyp"][0]
        s2 = g[tag + "_s2"] if tag + "_s2" in g.files else None
        cov_N = orc.cov_count(model["kernel"], D)
        noise_N = orc.noise_count(model["noise"])
        p = model["noise"]
        noise = gpr.noise_functions.GaussianNoise(p[0] == 1, p[1] >= 1, p[1] == 2, p[2] == 1)
        assert noise.hyperparameter_count() == noise_N
        a, da = noise.compute(hyp[cov_N:cov_N + noise_N], X, y, s2, compute_grad=True)
        b, db = orc.noise(p, hyp[cov_N:cov_N + noise_N], X, y, s2, compute_grad=True)
        assert np.isscalar(a) == np.isscalar(b) and np.array_equal(a, b) and np.array_equal(da, db)
        mean = means[model["mean"]]()
        m, dm = mean.compute(hyp[cov_N + noise_N:], X, compute_grad=True)
        rm, rdm = orc.mean(model["mean"], hyp[cov_N + noise_N:], X, compute_grad=True)
        assert np.array_equal(m, rm) and np.array_equal(np.asarray(dm), np.asarray(rdm))


def test_plugin_validation_messages():
    import gpyreg_amd as gpr

    X = np.ones((5, 2))
    with pytest.raises(ValueError) as e:
        gpr.mean_functions.ConstantMean().compute(np.ones(3), X)
    assert "Expected 1 mean function hyperparameters" in e.value.args[0]
    with pytest.raises(ValueError) as e:
        gpr.mean_functions.NegativeQuadratic().compute(np.ones((5, 1)), X)
    assert "Mean function output is available only for" in e.value.args[0]
    with pytest.raises(ValueError) as e:
        gpr.noise_functions.GaussianNoise(constant_add=True).compute(np.ones(2), X, None)
    assert "Expected 1 noise function hyperparameters" in e.value.args[0]
    with pytest.raises(ValueError) as e:
        gpr.noise_functions.GaussianNoise(constant_add=True).compute(np.ones((1, 1)), X, None)
    assert "Noise function output is available only for" in e.value.args[0]


def test_hyperparameter_dict_roundtrip_and_order():
    import gpyreg_amd as gpr

    gp = gpr.GP(3, gpr.covariance_functions.RationalQuadraticARD(), gpr.mean_functions.NegativeQuadratic(),
                gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True, scale_user_provided=True))
    n = 5 + 2 + 7
    arr = np.arange(2 * n, dtype=float).reshape(2, n)
    d = gp.hyperparameters_to_dict(arr)
    assert list(d[0]) == ["covariance_log_lengthscale", "covariance_log_outputscale", "covariance_log_shape",
                          "noise_log_scale", "noise_provided_log_multiplier", "mean_const", "mean_location",
                          "mean_log_scale"]
    assert np.array_equal(gp.hyperparameters_from_dict(d), arr)
    with pytest.raises(ValueError):
        gp.hyperparameters_to_dict(np.zeros((1, n + 1)))
    assert np.isnan(gp.get_hyperparameters(as_array=True)).all()
    gp.update(hyp=arr, compute_posterior=False)
    assert np.array_equal(gp.get_hyperparameters(as_array=True), arr)


def test_assemble_profiles_refuses_a_directory_with_two_runs(tmp_path):
    """tools/assemble_profiles.py must not pick one of several CSVs of a pass (round 2's "r02f" set mixed two code
    states that way): exactly one file per pass, or it stops."""
    import shutil
    import subprocess
    import sys

    work = tmp_path / "repo"
    (work / "gpurun_out" / "rX" / "prof" / "a").mkdir(parents=True)
    (work / "gpurun_out" / "rX" / "prof" / "b").mkdir(parents=True)
    (work / "profiles").mkdir()
    (work / "gpurun_out" / "rX" / "source.sha256").write_text("0" * 64)
    for d in ("a", "b"):
        (work / "gpurun_out" / "rX" / "prof" / d / f"1_kernel_stats.csv").write_text("Name,Calls\n")
    shutil.copy(os.path.join(ROOT, "tools", "assemble_profiles.py"), work / "assemble_profiles.py")
    r = subprocess.run([sys.executable, "assemble_profiles.py", "rX", "r99"], cwd=work, capture_output=True, text=True)
    assert r.returncode != 0 and "more than one run" in (r.stderr + r.stdout)
    assert not list((work / "profiles").iterdir())


def test_source_hash_changes_with_the_sources_only():
    import hashlib

    from tools.source_hash import source_hash

    h = source_hash()
    assert len(h) == 64 and h == source_hash()
    assert h != hashlib.sha256(b"").hexdigest()


def test_factor_of_a_covariance_that_lost_definiteness_matches_the_reference():
    """GP._robust_factor (the reference's __robust_cholesky, gaussian_process.py:2331-2355) on matrices LAPACK's
    Cholesky rejects -- rank-deficient, indefinite, an exact zero pivot -- against the reference's own output
    (tests/golden/draw_cases.npz), and on a positive definite one, where it is the upper Cholesky factor."""
    import os

    import numpy as np
    import scipy.linalg as sla

    from gpyreg_amd.gaussian_process import GP

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "draw_cases.npz"), allow_pickle=False)
    for k in range(int(g["n_rc"])):
        C, ref = g[f"rc{k}_C"], g[f"rc{k}_T"]
        T = GP._robust_factor(C.copy())
        assert T.shape == ref.shape, k
        assert np.allclose(T, ref, rtol=1e-9, atol=1e-12), k
    A = np.random.default_rng(5).standard_normal((6, 6))
    C = A @ A.T + 6 * np.eye(6)
    assert np.array_equal(GP._robust_factor(C), sla.cholesky(C))


def test_a_gp_without_device_state_pickles_on_the_cpu():
    """A stored GP must open on a machine without a GPU: pickling touches no device state."""
    import copy
    import pickle

    import numpy as np

    import gpyreg_amd as gpr

    gp = gpr.GP(3, gpr.covariance_functions.Matern(3), gpr.mean_functions.NegativeQuadratic(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.temporary_data["note"] = np.arange(3)
    for cp in (pickle.loads(pickle.dumps(gp)), copy.deepcopy(gp)):
        assert repr(cp).splitlines()[1] == repr(gp).splitlines()[1] == "    self.D = 3," and cp.covariance.degree == 3 and cp.posteriors is None and not cp._rebuild
        assert np.array_equal(cp.temporary_data["note"], np.arange(3))
        assert np.array_equal(cp.lower_bounds, gp.lower_bounds, equal_nan=True)


def test_repr_and_str_have_the_reference_layout():
    """test__str__and__repr__ of the reference (testing/test_gaussian_process.py:1031-1110) looks for these pieces."""
    import numpy as np

    import gpyreg_amd as gpr

    gp = gpr.GP(1, gpr.covariance_functions.Matern(3), gpr.mean_functions.ConstantMean(),
                gpr.noise_functions.GaussianNoise(constant_add=True))
    gp.set_bounds({"covariance_log_lengthscale": (-10.8, 3.0), "covariance_log_outputscale": (-5.0, 5.0),
                   "noise_log_scale": (-7.0, 2.0), "mean_const": (-3.0, 3.0)})
    assert "Covariance function: Matern" in str(gp)
    r = repr(gp)
    assert r.startswith("GP:\n    self.D = 1,\n    self.covariance = <gpyreg_amd.covariance_functions.Matern object at ")
    assert "self.lower_bounds = [-10.8" in r and "self.temporary_data = <dict object at " in r
    assert "_post_handle" not in r and "_token" not in r


def test_plugin_protocol_sweep_matches_the_reference_output():
    """tools/plugin_sweep.py (counts, info, get_bounds_info of every covariance / mean / noise class on four data sets
    including N = 1 and a duplicated point; values and gradients of every mean and of all 16 noise configurations, with
    and without user-provided noise) against the reference's output of the same script, to ten digits."""
    import contextlib
    import io
    import os
    import runpy
    import warnings

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    buf = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stdout(buf):
        warnings.simplefilter("ignore")
        runpy.run_path(os.path.join(root, "tools", "plugin_sweep.py"), run_name="__main__")
    mine = buf.getvalue().splitlines()
    with open(os.path.join(root, "tests", "golden", "plugin_sweep_reference.txt")) as f:
        ref = [ln.rstrip("\n") for ln in f]
    assert len(mine) == len(ref) > 700

    def same(a, b):
        ta, tb = a.split(), b.split()
        if len(ta) != len(tb):
            return False
        for x, y in zip(ta, tb):
            if x == y:
                continue
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                return False
            if abs(fx - fy) > 1e-9 * max(1.0, abs(fx), abs(fy)):
                return False
        return True

    bad = [(r, m) for r, m in zip(ref, mine) if not same(r, m)]
    assert not bad, "\n".join("reference: %s\nhere:      %s" % p for p in bad[:10])


def test_design_kernel_table_is_the_one_generated_from_the_committed_profiles():
    """DESIGN.md section 5's per-kernel table is generated (tools/design_kernel_table.py) from the rocprofv3 summaries
    under profiles/ so that it cannot go stale: the block between the markers must be exactly what
    the generator prints for the newest round tag that has a kernel-stats file."""
    import glob
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tags = sorted(re.match(r"(r\d+)_kernel_stats_bench_cfg3_groups1\.csv", os.path.basename(f)).group(1)
                  for f in glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_kernel_stats_bench_cfg3_groups1.csv")))
    assert tags, "no round-tagged kernel stats under profiles/"
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "design_kernel_table.py"), tags[-1]],
                         capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    design = open(os.path.join(root, "DESIGN.md")).read()
    m = re.search(r"<!-- kernel-table:begin -->.*?<!-- kernel-table:end -->", design, flags=re.S)
    assert m and m.group(0).strip() == out.stdout.strip(), "DESIGN.md section 5 is not the generator's output: run " \
        f"python tools/design_kernel_table.py {tags[-1]} --write"

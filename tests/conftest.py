"""pytest configuration: registers the ``gpu`` marker and puts the repo root on
sys.path so that ``oracle`` (test infrastructure) and ``gpyreg_amd`` import."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line(
        "markers", "experiments: needs the experiments build of the library (python -m gpyreg_amd.build --experiments, "
                   "then GPYREG_AMD_LIB=gpyreg_amd/lib/libgpcore_exp.so); skipped with the product library")


def pytest_collection_modifyitems(config, items):
    """Tests of the schedules that live in the experiments build only (dataflow graph, independent pipelines,
    rectangular tiles, right-looking panels) opt in: they run when GPYREG_AMD_LIB selects that build."""
    marked = [it for it in items if it.get_closest_marker("experiments")]
    if not marked:
        return
    try:
        from gpyreg_amd import _lib

        on = _lib.is_experiments_build()
    except Exception:  # noqa: BLE001 - no library at all: the ABI tests report that
        on = False
    if not on:
        skip = pytest.mark.skip(reason="experiments build only (GPYREG_AMD_LIB=gpyreg_amd/lib/libgpcore_exp.so)")
        for it in marked:
            it.add_marker(skip)


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def cov_golden():
    return _load("cov_cases.npz")


@pytest.fixture(scope="session")
def core_golden():
    return _load("core_cases.npz")


def parse_core_name(name):
    tag, kname, mname, npar, N, D, flavour = str(name).split("|")
    degree = 0
    kernel = kname
    for base in ("matern_iso", "matern"):
        if kname.startswith(base) and kname != base:
            kernel, degree = base, int(kname[len(base):])
            break
    model = dict(
        kernel=kernel, degree=degree, mean=mname, noise=tuple(int(c) for c in npar)
    )
    return tag, model, int(N), int(D), flavour


def parse_cov_name(name):
    tag, kname, N, D, M = str(name).split("|")
    degree = 0
    kernel = kname
    for base in ("matern_iso", "matern"):
        if kname.startswith(base) and kname != base:
            kernel, degree = base, int(kname[len(base):])
            break
    return tag, kernel, degree, int(N), int(D), int(M)

"""Seeded GP.fit over the option edge cases of the reference's tests (tools/fit_sweep.py: opts_N / n_samples / init_N
of 0 or 1 and their combinations in a row on one object, fixed and partly fixed bounds, user-provided noise, a second
fit on more data, recommended bounds) against the output of the REFERENCE running the same script
(tests/golden/fit_sweep_reference.txt).  The printed numbers carry four decimals; they are compared to 5e-3 (the
optimiser's tolerance, as in test_gpu_fit.py)."""

import contextlib
import io
import os
import runpy
import warnings

import pytest

from test_gpu_api_sweep import _tokens

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(a, b, tol):
    ta, tb = _tokens(a), _tokens(b)
    if len(ta) != len(tb):
        return False
    for x, y in zip(ta, tb):
        try:
            fx, fy = float(x), float(y)
        except ValueError:
            if x != y:
                return False
            continue
        if not (fx == fy or abs(fx - fy) <= tol * max(1.0, abs(fx), abs(fy))):
            return False
    return True


def test_fit_sweep_matches_the_reference_output():
    buf = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stdout(buf):
        warnings.simplefilter("ignore")
        runpy.run_path(os.path.join(ROOT, "tools", "fit_sweep.py"), run_name="__main__")
    mine = buf.getvalue().splitlines()
    with open(os.path.join(ROOT, "tests", "golden", "fit_sweep_reference.txt")) as f:
        ref = [ln.rstrip("\n") for ln in f]
    assert len(mine) == len(ref) >= 40, (len(mine), len(ref))
    assert sum("RAISES" in ln for ln in ref) == 2  # the two unknown samplers, by design
    bad = [(r, m) for r, m in zip(ref, mine) if not _close(r, m, 5e-3)]
    assert not bad, "\n".join("reference: %s\nhere:      %s" % p for p in bad[:10])

"""GP.update call sequences (tools/update_sweep.py) against the output of the REFERENCE running the same script
(tests/golden/update_sweep_reference.txt): one point / several points / new hyperparameters / compute_posterior off
and on / a GP that starts without data, for constant noise, noise below 1e-6, user-provided and output-dependent
noise, with a prediction after every call.  The reference itself fails in places (its rank-one path raises with
output-dependent noise and leaves the object inconsistent; ``predict`` after ``compute_posterior=False`` raises a
TypeError): a line on which the reference raises is not compared, and after a failed ``update`` of the reference the
rest of that object's sequence is skipped.  Where both raise by design (X without y, wrong dimension) both must."""

import contextlib
import io
import os
import runpy
import warnings

import pytest

from test_gpu_numerics_sweep import _line_ok

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("quirks", [False, True])
def test_update_sequences_match_the_reference_output(quirks):
    """quirks: ``GP(..., reference_quirks=True)`` -- one new point together with new hyperparameters is appended under
    the OLD samples like the reference does, and that line is compared too."""
    if quirks:
        os.environ["SWEEP_QUIRKS"] = "1"
    try:
        _check_update_sweep(quirks)
    finally:
        os.environ.pop("SWEEP_QUIRKS", None)


def _check_update_sweep(quirks):
    buf = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stdout(buf):
        warnings.simplefilter("ignore")
        runpy.run_path(os.path.join(ROOT, "tools", "update_sweep.py"), run_name="__main__")
    mine = buf.getvalue().splitlines()
    with open(os.path.join(ROOT, "tests", "golden", "update_sweep_reference.txt")) as f:
        ref = [ln.rstrip("\n") for ln in f]
    assert len(mine) == len(ref) == 64
    compared, bad, broken = 0, [], None
    for r, m in zip(ref, mine):
        name, tag = r.split()[0], r.split()[1]
        assert m.split()[:2] == [name, tag]
        if tag == "nodata":
            broken = None  # a fresh object
        if tag in ("x_only", "bad_dim"):
            assert "RAISES" in r and "RAISES" in m, (r, m)  # an input without its observation; a wrong dimension
            continue
        if broken == name:
            continue
        if tag == "one_newhyp" and "(6, 3)" in r and not quirks:
            # one new point TOGETHER with new hyperparameters: the reference takes its rank-one path, which never looks
            # at ``hyp`` (gaussian_process.py:738-746 does not test it) -- the new samples are silently dropped.  Here
            # the new hyperparameters are honoured (full recompute): two samples instead of the old three.
            # (With per-point noise the reference recomputes, and agrees: that model's line is compared below.)
            assert "(6, 2)" in m
            continue
        if "update RAISES" in r:
            broken = name
            continue
        if "RAISES" in r:
            assert "RAISES" in m, (r, m)  # predict on posteriors that were not computed: both refuse
            continue
        compared += 1
        if not _line_ok(r, m, 1e-7):
            bad.append((r, m))
    assert not bad, "\n".join("reference: %s\nhere:      %s" % p for p in bad[:10])
    assert compared == (38 if quirks else 36), compared  # (with quirks: + the one_newhyp lines of the two models that take the rank-one path); 12 + 12 + 11 of the three models the reference gets through, 1 of the fourth

"""Small-N single-evaluation latency, device vs the CPU oracle (printed with -s; the assertion is
only that the device result matches the oracle, the timing is a report)."""
import time

import numpy as np
import pytest

import bench
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [200, 1000])
def test_single_evaluation_latency_report(N):
    bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
    try:
        X, y, hyp = bench.synthetic_problem(2, 1)
        gp = bench.make_gp(2, "f64")
        gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
        model = dict(kernel="se", degree=0, mean="const", noise=(1, 0, 0))
        time.sleep(0.5)  # let the BLAS worker threads of an earlier oracle call stop spinning
        for _ in range(5):
            gp.nll_batch(hyp, True)
        t0 = time.perf_counter()
        for _ in range(20):
            nlz, dnlz = gp._GP__compute_nlZ(hyp[0], True, False)
        tg = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        ref, dref = orc.core(model, hyp[0], X, y, None, 1, 1)
        tc = time.perf_counter() - t0
        assert abs(nlz - ref) <= 1e-8 * max(1.0, abs(ref))
        assert np.allclose(dnlz, dref, rtol=1e-7, atol=1e-8 * np.abs(dref).max())
        print(f"\nN={N}: NLL+grad device {tg*1e3:.3f} ms, CPU oracle {tc*1e3:.1f} ms")
    finally:
        bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=2048)

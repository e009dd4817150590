"""leaf kernel duration vs number of blocks per CU (is the 49 us of a leaf instruction-fetch bound?
a CU that runs several leaves in a row has the code in its instruction cache from the second on)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpyreg_amd import _lib

N = 128
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=N)
for S in (16, 256, 1024, 4096):
    X, y, hyp = bench.synthetic_problem(2, S)
    gp = bench.make_gp(2, "f64")
    gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
    for _ in range(3):
        gp.nll_batch(hyp, False)
    print(f"S={S}: device {_lib.context().last_timing()[0]:.3f} ms", flush=True)

"""repeated GP.update (posterior sets) wall time with host phase breakdown."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys, time
import numpy as np
import bench
cfg, S = int(sys.argv[1]), int(sys.argv[2])
X, y, hyp = bench.synthetic_problem(cfg, S)
gp = bench.make_gp(cfg, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp[:1], compute_posterior=False)
for it in range(5):
    if it == 4:
        os.environ["GPC_HOSTTIME"] = "1"
    t0 = time.perf_counter(); gp.update(hyp=hyp); t = time.perf_counter() - t0
    print(f"update {it}: {t*1e3:.1f} ms", flush=True)

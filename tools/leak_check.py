import sys; sys.path.insert(0, ".")
import numpy as np, torch, bench, gc
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=700)
X, y, hyp = bench.synthetic_problem(2, 8)
gp = bench.make_gp(2, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp)
xs = X[:50] + 0.01
def free(): 
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
for rep in range(4):
    for _ in range(100):
        gp.update(hyp=hyp + 0.01 * np.random.randn(*hyp.shape))
        gp.predict(xs)
        gp.nll_batch(hyp, True)
    gc.collect()
    print("free MiB after", (rep + 1) * 100, "cycles:", round(free(), 1), flush=True)

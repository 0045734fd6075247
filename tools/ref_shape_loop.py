"""The reference's own timed call pattern (bench.py reference_test_shape leg) as a bare loop, for traces:
376x240 farmsim pair, PM_SEM_GPU, alpha 0.9, 3 iterations, self-seeded, host in / host out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import pm_ctypes as pm
pm.load()
g = np.load(os.path.join(ROOT, "tests", "golden", "farmsim_fs1_376x240.npz"))
l, r = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
rows, cols = l.shape
prm = pm.default_params(pm.PM_SEM_GPU, cost_alpha=0.9, patchmatch_iters=3, sparse_init=1)
with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
    out = (np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32))
    ts = []
    for i in range(int(os.environ.get("CALLS", "30"))):
        t0 = time.perf_counter()
        e.match(l, r, out=out)
        ts.append(1e3 * (time.perf_counter() - t0))
print("median ms per call", round(float(np.median(ts[5:])), 3))

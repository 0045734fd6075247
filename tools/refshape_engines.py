import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import pm_ctypes as pm
pm.load()
g = np.load(os.path.join(ROOT, "tests", "golden", "farmsim_fs1_376x240.npz"))
l, r = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
rows, cols = l.shape
ref = None
for eng in (0, 2, 1):
    prm = pm.default_params(pm.PM_SEM_GPU, cost_alpha=0.9, patchmatch_iters=3, sparse_init=1, engine=eng)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        out = (np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32))
        ts = []
        for i in range(40):
            t0 = time.perf_counter()
            e.match(l, r, out=out)
            ts.append(1e3 * (time.perf_counter() - t0))
        if ref is None: ref = (out[0].copy(), out[1].copy())
        print("engine", eng, "median ms per call", round(float(np.median(ts[5:])), 3), "same", np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]), flush=True)

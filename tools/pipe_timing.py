import os, sys, time
sys.path.insert(0, "/root/repo/ocean-perception_amd/python")
import numpy as np
import pm_ctypes as pm, synth
pm.load()
rows, cols = 720, 1280
p = synth.make_pair(0, rows, cols)
seeded = len(sys.argv) < 2 or sys.argv[1] != "noseed"
seeds = (p["seed_l"], p["seed_r"]) if seeded else (None, None)
prm = pm.default_params(0, patch=11, patchmatch_iters=8)
bufs = [(np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)) for _ in range(4)]
with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=3) as e:
    e.match(p["left"], p["right"], *seeds, out=bufs[0])
    ts, tc, n, k = 0.0, 0.0, int(os.environ.get("N", "60")), 0
    t0 = time.perf_counter()
    for i in range(n):
        if e.in_flight() == 3:
            a = time.perf_counter(); e.collect(out=bufs[k & 3]); tc += time.perf_counter() - a; k += 1
        a = time.perf_counter(); e.submit(p["left"], p["right"], *seeds, tag=i); ts += time.perf_counter() - a
    while e.in_flight():
        a = time.perf_counter(); e.collect(out=bufs[k & 3]); tc += time.perf_counter() - a; k += 1
    dt = time.perf_counter() - t0
    print(f"seeded {seeded}: {n / dt:.1f} pairs/s; per frame: total {1e3 * dt / n:.3f} ms, in submit {1e3 * ts / n:.3f} ms, in collect {1e3 * tc / n:.3f} ms")

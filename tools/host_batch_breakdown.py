import os, sys, time
sys.path.insert(0, "/root/repo/ocean-perception_amd/python")
import numpy as np
import pm_ctypes as pm, synth
pm.load()
rows, cols, nb = 720, 1280, 4
prs = [synth.make_pair(i, rows, cols) for i in range(nb)]
prm = pm.default_params(0, patch=11, patchmatch_iters=8)
with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=nb) as e:
    args = ([p["left"] for p in prs], [p["right"] for p in prs], [p["seed_l"] for p in prs], [p["seed_r"] for p in prs])
    e.match_batch(*args)
    for rep in range(3):
        t0 = time.perf_counter(); e.match_batch(*args); t1 = time.perf_counter()
        print(f"match_batch(4): {1e3*(t1-t0):.2f} ms")
    bufs = [(np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)) for _ in range(4)]
    for rep in range(3):
        t0 = time.perf_counter(); k = 0
        for i in range(nb):
            if e.in_flight() == 3:
                e.collect(out=bufs[k]); k += 1
            e.submit(prs[i]["left"], prs[i]["right"], prs[i]["seed_l"], prs[i]["seed_r"], tag=i)
        while e.in_flight():
            e.collect(out=bufs[k]); k += 1
        t1 = time.perf_counter()
        print(f"manual submit/collect of 4, reused outputs: {1e3*(t1-t0):.2f} ms")
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(nb):
            e.match(prs[i]["left"], prs[i]["right"], prs[i]["seed_l"], prs[i]["seed_r"], out=bufs[i])
        t1 = time.perf_counter()
        print(f"4 x match(): {1e3*(t1-t0):.2f} ms")

"""pm_host_alloc memory (owned by the handle) through the frame sequence; prints what fails instead of raising."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import numpy as np
import pm_ctypes as pm, synth
pm.load()
rows, cols = 60, 100
pairs = [synth.make_pair(140 + i, rows=rows, cols=cols, n_points=25, dilate_factor=2) for i in range(9)]
params = pm.default_params(0, patch=5, patchmatch_iters=2)
with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
    want = [e.match(p["left"], p["right"], p["seed_l"], p["seed_r"]) for p in pairs]
e = pm.Engine(params, max_rows=rows + 4, max_cols=cols + 8, max_batch=4)
try:
    ins = []
    for p in pairs:
        row = []
        for k in ("left", "right", "seed_l", "seed_r"):
            b = e.host_alloc(p[k].shape, p[k].dtype, owned=True)
            np.copyto(b, p[k])
            row.append(b)
        ins.append(row)
    outs = [(e.host_alloc((rows, cols), np.float32, owned=True), e.host_alloc((rows, cols), np.float32, owned=True)) for _ in range(4)]
    got = []
    for i in range(len(pairs)):
        if e.in_flight() == 4:
            dl, dr, tag = e.collect()
            got.append((dl.copy(), dr.copy()))
        e.submit(*ins[i], tag=i, out=outs[i % 4])
    while e.in_flight():
        dl, dr, tag = e.collect()
        got.append((dl.copy(), dr.copy()))
    bad = [i for i in range(len(pairs)) if not (np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1]))]
    print("owned memory sequence: mismatching frames", bad)
    e.host_free(outs[0][0])
    print("host_free ok")
except Exception as ex:
    print("FAILED:", repr(ex))
    traceback.print_exc(limit=3)
finally:
    del ins, outs
    e.close()
print("done")

"""Seeder timing: pm_sparse_init on resident 1280x720 images, per-kernel from rocprofv3 or whole-call here.

    python tools/bench_seed.py [--reps 50]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pm_ctypes as pm
import synth

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
pm.load()
p = synth.make_pair(0, 720, 1280)
with pm.Engine(pm.default_params(1), max_rows=720, max_cols=1280) as e:
    for _ in range(3):
        e.sparse_init(p["left"], p["right"], 4)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        e.sparse_init(p["left"], p["right"], 4)
    dt = (time.perf_counter() - t0) / a.reps
print(f"pm_sparse_init host-to-host: {dt * 1e3:.3f} ms per view (includes the PCIe copies)")

"""Differential fuzz of the self-seeded Match() -- the reference's own call pattern, PatchmatchGpu::Match(iml, imr,
disp, dispr) with sparse_init on (patchmatch_gpu.cu:331-376): device SparseInit on both views (view 1 on the mirrored
pair), then the iterations -- against the same composition of the oracles, for both scalar semantics and the plane
mode, random sizes, detector / matcher parameters, dilate factors and explicit-seed overrides.

    python tools/fuzz_selfseed.py [--cases 40] [--seed 1]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pm_ctypes as pm
import oracle_lib as oracle
import synth
from test_planes import okw

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
pm.load()
oracle.load()
rng = np.random.default_rng(a.seed)
t0 = time.time()
for case in range(a.cases):
    which = int(rng.integers(0, 3))   # 0: PM_SEM_CPU, 1: PM_SEM_GPU, 2: planes
    rows = int(rng.integers(60, 200))
    cols = int(rng.integers(150, 420))
    iters = int(rng.integers(1, 4))
    f = int(rng.integers(1, 5))
    maxf = int(rng.choice([20, 200, 500]))
    mind = int(rng.choice([3, 10, 20]))
    md = int(rng.choice([64, 128]))
    patch = int(rng.choice([3, 5, 11])) if which != 1 else 3
    skw = dict(max_features_per_frame=maxf, min_distance_btw_features=mind, max_disp=md, sparse_init=1,
               init_dilate_factor=f)
    sp = oracle.seed_params(max_features=maxf, min_distance=mind, max_disp=md)
    p = synth.make_pair(int(rng.integers(0, 1000)), rows=rows, cols=cols)
    l, r = p["left"], p["right"]
    explicit_left = bool(rng.random() < 0.3)   # an explicit left seed map takes precedence over the device seeder
    osl = p["seed_l"] if explicit_left else oracle.sparse_init(l, r, f, sp)
    osr = np.ascontiguousarray(oracle.sparse_init(r[:, ::-1], l[:, ::-1], f, sp)[:, ::-1])
    if which == 2:
        prm = pm.default_params(0, patch=patch, patchmatch_iters=iters, mode=pm.PM_MODE_PLANES,
                                state_dtype=int(rng.integers(0, 2)), **skw)
        want = oracle.planes_match(oracle.planes_params(**okw(prm)), l, r, osl, osr)
    else:
        prm = pm.default_params(which, patch=patch, patchmatch_iters=iters, **skw)
        want = oracle.match(oracle.default_params(which, patch=patch, n_iters=iters, nthreads=8), l, r, osl, osr)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        got = e.match(l, r, p["seed_l"] if explicit_left else None, None)
    ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    print(f"case {case:3d}: {['PM_SEM_CPU', 'PM_SEM_GPU', 'planes'][which]} {cols}x{rows} patch {patch} iters {iters} f {f} "
          f"maxf {maxf} mind {mind} max_disp {md} explicit_left {int(explicit_left)} "
          f"{'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
    if not ok:
        sys.exit(1)
print("all", a.cases, "cases bit-identical")

"""Differential fuzz of the sweep engines on the GPU: the product engine (PM_ENGINE_RUNBLK2, every group-width
heuristic) against the one-lane-per-chain serial anchor, whole Match() with both views, random sizes / windows /
iteration counts / noise schedules / seed densities.  Bit-exact or it prints the case and exits 1.

    python tools/fuzz_engines.py [--cases 60] [--seed 1]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pm_ctypes as pm
import synth

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--big", action="store_true", help="sizes up to 2600 x 1800: chains beyond 1600 positions (8 wavefronts per chain),\n"
                "column sweeps whose reference lines no longer fit the LDS budget")
ap.add_argument("--gpu-share", type=float, default=0.2, help="share of PM_SEM_GPU cases")
a = ap.parse_args()
pm.load()
rng = np.random.default_rng(a.seed)
t0 = time.time()
for case in range(a.cases):
    sem = 1 if rng.random() < a.gpu_share else 0
    patch = int(rng.choice([3, 5, 7, 9, 11])) if sem == 0 else 3
    rows = int(rng.integers(2 * patch + 8, 260))
    cols = int(rng.integers(2 * patch + 40, 700))
    if a.big:
        rows, cols = int(rng.integers(300, 1800)), int(rng.integers(500, 2600))
    iters = int(rng.integers(1, 9))
    amp0 = float(rng.choice([32.0, 8.0, 2.0, 0.75]))
    noise = [amp0 / (2 ** i) for i in range(iters)]
    p = synth.make_pair(int(rng.integers(0, 1000)), rows=rows, cols=cols, n_points=int(rng.integers(3, 60)),
                        dilate_factor=int(rng.integers(1, 4)))
    sl, sr = p["seed_l"].copy(), p["seed_r"].copy()
    kind = rng.integers(0, 4)
    if kind == 1:      # near-integer sample positions: the lerp weights at their extremes
        xs = np.arange(cols, dtype=np.float32)[None, :]
        k = rng.integers(0, 40, (rows, cols)).astype(np.float32)
        eps = rng.choice(np.array([0, 2.0 ** -17, -2.0 ** -17, 7.7e-6, -7.7e-6, 0.5], np.float32), (rows, cols))
        sl = np.maximum(xs - np.float32((patch - 1) * 0.5) - k - eps, 0).astype(np.float32)
    elif kind == 2:    # long plateaus
        sl = np.repeat(np.repeat(rng.uniform(0, 40, (rows // 7 + 1, cols // 23 + 1)), 7, 0), 23, 1)[:rows, :cols].astype(np.float32)
    elif kind == 3:    # sparse seeds
        sl[rng.random((rows, cols)) < 0.9] = 0
    outs = []
    for engine in (1, 5):
        prm = pm.default_params(sem, patch=patch, patchmatch_iters=iters, engine=engine, noise_amp=noise)
        with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
            outs.append(e.match(p["left"], p["right"], sl, sr))
    ok = all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))
    print(f"case {case:3d}: sem {sem} {cols}x{rows} patch {patch} iters {iters} amp0 {amp0} kind {kind} "
          f"{'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
    if not ok:
        bad = np.argwhere(outs[0][0] != outs[1][0])
        print("first left mismatch at", bad[:3].tolist() if len(bad) else None)
        sys.exit(1)
print("all", a.cases, "cases bit-identical")

"""Differential fuzz of the row-tiled protocol (python/tiled.py + pm_tile_*; ranks as threads of one process on one
GPU): random sizes, band counts, windows, iteration counts, exchange rounds (0 forces the repeat path), both
semantics -- tiled == untiled bit for bit, or it prints the case and exits 1.  Every case also goes through the C-ABI
driver (pm_tiled_*, csrc/pm_tiled.hip) with a random exchange mode and the bands accounted to random LOGICAL devices
(include/pm/testing.h): the maps must be the same and the runtime log must not hold one call that would be an error --
or a silent cross-device access -- with one band per GPU.

    python tools/fuzz_tiled.py [--cases 40] [--seed 1]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pm_ctypes as pm
import synth
import tiled

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
pm.load()
rng = np.random.default_rng(a.seed)
t0 = time.time()
for case in range(a.cases):
    sem = 0 if rng.random() < 0.8 else 1
    patch = int(rng.choice([3, 5, 7, 11])) if sem == 0 else 3
    world = int(rng.integers(2, 7))
    rows = int(rng.integers(world * (patch + 8), 320))
    cols = int(rng.integers(2 * patch + 40, 500))
    iters = int(rng.integers(1, 6))
    rounds = int(rng.choice([0, 1, 2, 2, 3]))
    p = synth.make_pair(int(rng.integers(0, 1000)), rows=rows, cols=cols, n_points=int(rng.integers(5, 80)),
                        dilate_factor=int(rng.integers(1, 4)))
    sl, sr = p["seed_l"], p["seed_r"]
    if rng.random() < 0.3:   # a few long vertical structures: values that cross many band boundaries in one sweep
        sl = sl.copy()
        for _ in range(6):
            x = int(rng.integers(patch, cols - patch))
            sl[:, x:x + 3] = np.float32(rng.uniform(2, 30))
    params = pm.default_params(sem, patch=patch, patchmatch_iters=iters)
    py_pipe = bool(rng.integers(0, 2))  # the Python driver's schedule: ranks in order, or speculative rounds
    dl, dr, info = tiled.match_tiled_local(params, p["left"], p["right"], sl, sr, world, rounds=rounds, pipelined=py_pipe)
    with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
        ul, ur = e.match(p["left"], p["right"], sl, sr)
    ok = np.array_equal(dl, ul) and np.array_equal(dr, ur)
    # the same pair through pm_tiled_*: bands on random logical devices, random exchange mode, random peer links
    ndev = int(rng.integers(1, world + 1))
    logical = sorted(int(v) for v in rng.integers(0, ndev, world))  # neighbours may share a device or not
    if rng.random() < 0.3:
        logical = logical[::-1]
    mode, peer, sched = int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    with pm.TiledEngine(params, rows, cols, world, logical_devices=logical, simulate_peer_access=peer, exchange=mode,
                        schedule=sched) as t:
        cl, cr, cinfo = t.match(p["left"], p["right"], sl, sr, rounds=rounds if rounds > 0 else 0)
        _, bad = t.audit()
    ok = ok and np.array_equal(cl, ul) and np.array_equal(cr, ur) and bad == 0
    print(f"case {case:3d}: sem {sem} {cols}x{rows} patch {patch} iters {iters} bands {world} rounds {rounds} "
          f"repeated {int(bool(info['repeated']))} pipelined {int(py_pipe)} | C driver: devices {logical} exchange {mode} peer {peer} schedule {sched} "
          f"marked calls {bad} {'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
    if not ok:
        sys.exit(1)
print("all", a.cases, "cases bit-identical")

"""Differential fuzz of the plane mode: HIP kernels (through the C ABI) against the CPU definition
(oracle/pm_planes_oracle.c), whole Match(), random sizes / windows / iteration counts / disparity ranges / slope and
schedule constants / seeds / f32 and f16 state.  Bit-exact or it prints the case and exits 1.

    python tools/fuzz_planes.py [--cases 40] [--seed 1]
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
    patch = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
    rows = int(rng.integers(patch + 6, 150))
    cols = int(rng.integers(patch + 30, 330))
    iters = int(rng.integers(1, 5))
    max_disp = int(rng.choice([8, 16, 48, 64, 128, 200]))
    f16 = int(rng.integers(0, 2))
    lr = int(rng.integers(0, 2))
    amp0 = float(rng.choice([32.0, 8.0, 3.0, 0.5]))
    kw = dict(max_disp=max_disp, left_right_check=lr, noise_amp=[amp0 / (2 ** i) for i in range(16)],
              plane_refine_steps=int(rng.integers(1, 6)), plane_slope_max=float(rng.choice([0.25, 0.5, 1.0])),
              plane_slope_init=float(rng.choice([0.0, 0.25, 0.5])), plane_slope_per_disp=float(rng.choice([1 / 64, 0.03])),
              plane_lr_tol=float(rng.choice([0.5, 1.0, 2.0])), noise_seed=int(rng.integers(1, 1 << 30)),
              plane_window=int(rng.integers(0, 2)), plane_neighbours=int(rng.integers(0, 2)))
    kw["plane_slope_init"] = min(kw["plane_slope_init"], kw["plane_slope_max"])
    prm = pm.default_params(0, patch=patch, patchmatch_iters=iters, mode=pm.PM_MODE_PLANES, state_dtype=f16, **kw)
    p = synth.make_pair(int(rng.integers(0, 1000)), rows=rows, cols=cols, n_points=int(rng.integers(3, 40)),
                        dilate_factor=int(rng.integers(1, 4)))
    seeded = rng.random() < 0.4
    sl, sr = (p["seed_l"], p["seed_r"]) if seeded else (None, None)
    want = oracle.planes_match(oracle.planes_params(**okw(prm)), p["left"], p["right"], sl, sr)
    # a third of the cases: the pair sits in a random slot of a batch of 2-5 pairs (the batch runs as two lanes on two
    # streams: pm_planes_host.hip::planes_match); its maps must be the same
    nb = int(rng.integers(2, 6)) if rng.random() < 0.34 else 1
    slot = int(rng.integers(0, nb))
    with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=nb) as e:
        if nb == 1:
            got = e.match(p["left"], p["right"], sl, sr)
        else:
            others = [synth.make_pair(int(rng.integers(0, 1000)), rows=rows, cols=cols, n_points=10, dilate_factor=2) for _ in range(nb)]
            others[slot] = p
            dls, drs = e.match_batch([q["left"] for q in others], [q["right"] for q in others],
                                     [q["seed_l"] for q in others] if seeded else None, [q["seed_r"] for q in others] if seeded else None)
            got = (dls[slot], drs[slot] if lr else None)
    ok = np.array_equal(got[0], want[0]) and (not lr or np.array_equal(got[1], want[1]))
    print(f"case {case:3d}: {cols}x{rows} patch {patch} window {kw['plane_window']} neigh {kw['plane_neighbours']} iters {iters} max_disp {max_disp} f16 {f16} lr {lr} seeded {int(seeded)} batch {nb}/{slot} "
          f"{'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
    if not ok:
        print("params:", kw)
        sys.exit(1)
print("all", a.cases, "cases bit-identical")

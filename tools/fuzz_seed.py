"""Differential fuzz of the device seeder (SparseInit: corner response, NMS, chunked threshold / counting sort /
greedy selection in one workgroup, template match, splat) against oracle/pm_seed_oracle.c: random sizes, image kinds
(synthetic scenes, white noise, periodic patterns with many IDENTICAL responses, flat regions), detector and matcher
parameters.  Bit-exact or it prints the case and exits 1.

    python tools/fuzz_seed.py [--cases 60] [--seed 1]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pm_ctypes as pm
import oracle_lib as oracle
import synth

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--only", type=int, default=-1, help="run just this case of the sequence (reproduce a failure)")
ap.add_argument("--dump", action="store_true", help="with --only: print where the maps differ")
a = ap.parse_args()
pm.load()
oracle.load()
rng = np.random.default_rng(a.seed)
t0 = time.time()
for case in range(a.cases):
    rows = int(rng.integers(40, 420))
    cols = int(rng.integers(140, 900))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        p = synth.make_pair(int(rng.integers(0, 1000)), rows=rows, cols=cols)
        left, right = p["left"], p["right"]
    elif kind == 1:
        left = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        right = np.roll(left, -int(rng.integers(1, 40)), axis=1)
    elif kind == 2:  # periodic: thousands of identical corner responses
        py, px = int(rng.integers(3, 17)), int(rng.integers(3, 17))
        tile = rng.integers(0, 256, (py, px), dtype=np.uint8)
        left = np.tile(tile, (rows // py + 1, cols // px + 1))[:rows, :cols].copy()
        right = np.roll(left, -int(rng.integers(1, 20)), axis=1)
    else:            # mostly flat with a few blobs
        left = np.full((rows, cols), 90, np.uint8)
        for _ in range(int(rng.integers(1, 30))):
            y, x = int(rng.integers(0, rows - 8)), int(rng.integers(0, cols - 8))
            left[y:y + int(rng.integers(2, 8)), x:x + int(rng.integers(2, 8))] = int(rng.integers(0, 256))
        right = np.roll(left, -int(rng.integers(1, 30)), axis=1)
    maxf = int(rng.choice([1, 7, 50, 200, 600, 1024]))
    mind = int(rng.choice([1, 2, 5, 20, 45]))
    q = float(rng.choice([0.001, 0.01, 0.1, 0.4]))
    blk = int(rng.choice([3, 5, 7, 9]))
    tc = int(rng.choice([11, 21, 31]))
    tr = int(rng.choice([5, 11]))
    md = int(rng.choice([64, 100, 128]))
    f = int(rng.integers(1, 5))
    # the detector / matcher options of round 4: Harris response, cv::cornerSubPix on corners and on matches
    harris = int(rng.random() < 0.3)
    hk = float(rng.choice([0.0, 0.04, 0.15]))
    spc = int(rng.random() < 0.35)
    spr = int(rng.random() < 0.25)
    swin, szero = int(rng.choice([2, 5, 10, 15])), int(rng.choice([-1, -1, 0, 1]))
    sit, seps = int(rng.choice([1, 5, 10, 40])), float(rng.choice([0.0, 0.001, 0.01, 0.1]))
    if spc and maxf > 200:   # one lane per corner, strictly sequential: keep the fuzz cases short
        maxf = 200
    prm = pm.default_params(1, max_features_per_frame=maxf, min_distance_btw_features=mind, gftt_quality_level=q,
                            gftt_block_size=blk, templ_cols=tc, templ_rows=tr, max_disp=md, gftt_use_harris=harris,
                            gftt_k=hk, subpixel_corners=spc, subpix_winsize=swin, subpix_zerozone=szero,
                            subpix_maxiters=sit, subpix_epsilon=seps, subpixel_refinement=spr)
    sp = oracle.seed_params(max_features=maxf, min_distance=mind, quality_level=q, block_size=blk, templ_cols=tc,
                            templ_rows=tr, max_disp=md, use_harris=harris, harris_k=hk, subpixel_corners=spc,
                            subpix_winsize=swin, subpix_zerozone=szero, subpix_maxiters=sit, subpix_epsilon=seps,
                            subpixel_refinement=spr)
    if a.only >= 0 and case != a.only:
        continue
    want = oracle.sparse_init(left, right, f, sp)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        got = e.sparse_init(left, right, f)
    ok = np.array_equal(got, want)
    print(f"case {case:3d}: {cols}x{rows} kind {kind} maxf {maxf} mind {mind} q {q} block {blk} templ {tc}x{tr} "
          f"max_disp {md} f {f} harris {harris} subpix {spc}{spr} win {swin} seeds {(want > 0).mean():.3f} {'ok' if ok else 'MISMATCH'}  [{time.time() - t0:.0f} s]",
          flush=True)
    if not ok:
        if a.dump:
            bad = np.argwhere(got != want)
            print(len(bad), "pixels differ; first:", bad[:5].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
            ys, xs = np.nonzero(want != got)
            print("bbox rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
        sys.exit(1)
print("all", a.cases, "cases bit-identical")

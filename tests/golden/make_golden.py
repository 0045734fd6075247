"""Regenerates the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py            # synthetic cases only
    python tests/golden/make_golden.py --caddy    # also the CADDY 640x480 fixture (needs the
                                                  # reference tree's test/resources, this container only)

Fixtures are data: inputs + expected outputs.  The CADDY fixture holds the two JPEGs of the
reference's test/resources (caddy_32_{left,right}.jpg, used by its sgbm/feature tests and named by
BASELINE.json configs[0]) decoded ONCE to 8-bit gray -- the luma plane as the JPEG library delivers it
(JCS_GRAYSCALE, what cv::imread(path, GRAYSCALE) requests; Pillow's libjpeg-turbo via Image.draft("L"),
bit-identical to the build's own decoder host/jpeg.cpp, tests/test_dataset.py) -- plus an
explicit seed map (block-matched on a sparse grid below; the reference's GFTT seeder needs OpenCV)
and per-row checksums of the oracle's disparity map for the 7x7 / 3-iteration configuration.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))

import oracle_lib as O  # noqa: E402
import synth  # noqa: E402


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
    print("wrote", name)


def synthetic():
    p = synth.make_pair(100, rows=48, cols=64, n_points=24, dilate_factor=2)
    for name, sem, patch in (("synth64x48_cpu5", 0, 5), ("synth64x48_gpu", 1, 3)):
        prm = O.default_params(sem, patch=patch, n_iters=3, nthreads=8, literal=1 if sem == 0 else 0)
        dl, dr = O.match(prm, p["left"], p["right"], p["seed_l"], p["seed_r"])
        save(name, left=p["left"], right=p["right"], seed_l=p["seed_l"], seed_r=p["seed_r"], disp_l=dl, disp_r=dr,
             sem=sem, patch=patch, iters=3, lr=1)
    # the recipe of test/stereo_matching/patchmatch_test.cpp:173-183 on a 150x96 pair (the size of the
    # reference's fsl3/fsr3 images)
    p = synth.make_pair(101, rows=96, cols=150, n_points=40, dilate_factor=2)
    sched = dict(noise_amp=[32.0, 8.0, 2.0, 0.5], patch_w=[5, 5, 3, 3], patch_h=[5, 5, 3, 3])
    prm = O.default_params(0, n_iters=4, bg_patch_w=3, bg_patch_h=3, bg_factor=1.5, nthreads=8, literal=1, **sched)
    dl, dr = O.match(prm, p["left"], p["right"], p["seed_l"], p["seed_r"])
    save("synth96x150_cpu_recipe", left=p["left"], right=p["right"], seed_l=p["seed_l"], seed_r=p["seed_r"],
         disp_l=dl, disp_r=dr, sem=0, iters=4, lr=1, bg_patch=3, **{k: np.array(v) for k, v in sched.items()})


def planes():
    # PM_MODE_PLANES has no reference counterpart: the fixture pins this build's own definition
    # (oracle/pm_planes_oracle.c) against regressions, for f32 and f16 state
    p = synth.make_pair(102, rows=64, cols=96, d_max=20.0)
    out = dict(left=p["left"], right=p["right"], patch=7, iters=3, max_disp=24)
    for f16 in (0, 1):
        prm = O.planes_params(n_iters=3, nthreads=8, state_f16=f16, patch=7, max_disp=24)
        dl, dr = O.planes_match(prm, p["left"], p["right"])
        out[f"disp_l_f{16 if f16 else 32}"] = dl
        out[f"disp_r_f{16 if f16 else 32}"] = dr
    save("planes_64x96", **out)


def block_match_seeds(left, right, n=200, tw=31, th=11, max_disp=98, dilate_factor=4):
    """Sparse SAD block matching on a regular grid -> dilated seed map (stand-in for SparseInit)."""
    rows, cols = left.shape
    gy = int(round(np.sqrt(n * rows / cols)))
    gx = int(round(n / gy))
    seed = np.zeros((rows, cols), np.float32)
    L, R = left.astype(np.int32), right.astype(np.int32)
    for iy in range(gy):
        for ix in range(gx):
            y = int((iy + 0.5) * rows / gy)
            x = int((ix + 0.5) * cols / gx)
            y0, x0 = y - th // 2, x - tw // 2
            if y0 < 0 or y0 + th > rows or x0 - max_disp < 0 or x0 + tw > cols:
                continue
            t = L[y0:y0 + th, x0:x0 + tw]
            if t.std() < 6:
                continue
            sad = [np.abs(t - R[y0:y0 + th, x0 - d:x0 - d + tw]).mean() for d in range(1, max_disp + 1)]
            d = int(np.argmin(sad)) + 1
            if sad[d - 1] < 12:
                seed[y, x] = d
    k = int(2 ** dilate_factor) + 1
    return O.dilate_rect(seed, k)


def caddy():
    from PIL import Image
    res = "/root/reference/test/resources"
    def luma(name):
        im = Image.open(os.path.join(res, name))
        im.draft("L", im.size)
        return np.asarray(im, dtype=np.uint8)

    left, right = luma("caddy_32_left.jpg"), luma("caddy_32_right.jpg")
    assert left.shape == (480, 640), left.shape
    seed_l = block_match_seeds(left, right)
    seed_r = np.zeros_like(seed_l)
    prm = O.default_params(0, patch=7, n_iters=3, nthreads=8, left_right_check=0)
    dl, _ = O.match(prm, left, right, seed_l, None)
    rows_sum = dl.view(np.uint32).astype(np.uint64).sum(axis=1)
    save("caddy_32_gray_640x480", left=left, right=right, seed_l=seed_l, seed_r=seed_r, checksum_rows=rows_sum,
         checksum_total=np.uint64(rows_sum.sum()), fg_fraction=np.float32((dl > 0).mean()))
    print("caddy: seed coverage %.3f, foreground after match %.3f" % ((seed_l > 0).mean(), (dl > 0).mean()))


if __name__ == "__main__":
    if "--planes-only" in sys.argv:
        planes()
        sys.exit(0)
    synthetic()
    planes()
    if "--caddy" in sys.argv:
        caddy()

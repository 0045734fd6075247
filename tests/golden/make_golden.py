"""Regenerates the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py            # synthetic cases only
    python tests/golden/make_golden.py --caddy    # also the CADDY 640x480 fixture (needs the
                                                  # reference tree's test/resources, this container only)
    python tests/golden/make_golden.py --farmsim  # also the fsl1 / fsr1 pair of the reference's two PatchMatch tests

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
    # both windows of the definition: the checkerboard (default since round 4) and the full one ("full_" keys)
    for f16 in (0, 1):
        for window, tag in ((O.PL_WINDOW_CHECKER, ""), (O.PL_WINDOW_FULL, "full_")):
            prm = O.planes_params(n_iters=3, nthreads=8, state_f16=f16, patch=7, max_disp=24, window=window)
            dl, dr = O.planes_match(prm, p["left"], p["right"])
            out[f"{tag}disp_l_f{16 if f16 else 32}"] = dl
            out[f"{tag}disp_r_f{16 if f16 else 32}"] = dr
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


def png_rgb_to_gray(rgb):
    """8-bit RGB -> gray as cv::imread(path, IMREAD_GRAYSCALE) gets it for a PNG file: OpenCV 3.4's PngDecoder asks
    libpng for the conversion (png_set_rgb_to_gray(png_ptr, 1, 0.299, 0.587)), and libpng's png_do_rgb_to_gray computes
    (rc * R + gc * G + bc * B + 16384) >> 15 with rc = 0.299 * 32768 = 9798 (rounded), gc = 19235, bc = 32768 - rc - gc.
    Restated from the published algorithm; parity with the reference's binary stays unpinned."""
    rc, gc = 9798, 19235
    bc = 32768 - rc - gc
    v = rgb.astype(np.int64)
    g = (rc * v[..., 0] + gc * v[..., 1] + bc * v[..., 2] + 16384) >> 15
    same = (rgb[..., 0] == rgb[..., 1]) & (rgb[..., 1] == rgb[..., 2])
    return np.where(same, rgb[..., 0], g).astype(np.uint8)


def farmsim():
    """The inputs of BOTH reference PatchMatch tests (test/stereo_matching/patchmatch_test.cpp:121-133,
    patchmatch_gpu_test.cpp:52-64): fsl1.png / fsr1.png, read as gray, halved with cv::resize (INTER_LINEAR) to
    376x240.  The reference asserts nothing on them and ships no expected maps, so the fixture pins what this build's
    restatement computes for the two recipes: per-row checksums of
      (a) patchmatch_test.cpp:149-183 -- Initialize(il, ir, 1), noise 32 / 8 / 2 / 0.5 with 5x5, 5x5, 3x3, 3x3,
          RemoveBackground(3x3, 1.5), one view;
      (b) patchmatch_gpu_test.cpp:68-88 -- PatchmatchGpu with cost_alpha 0.9, 3 iterations, self-seeded by SparseInit(4),
          both views + MaskOcclusions (Match is called five times on the same pair: the same result five times)."""
    from PIL import Image
    res = "/root/reference/test/resources/images"
    gray = {}
    for side, name in (("left", "fsl1.png"), ("right", "fsr1.png")):
        rgb = np.asarray(Image.open(os.path.join(res, name)).convert("RGB"), dtype=np.uint8)
        assert rgb.shape == (480, 752, 3), rgb.shape
        gray[side] = O.resize_linear_u8(png_rgb_to_gray(rgb), 240, 376)
    l, r = gray["left"], gray["right"]
    rowsum = lambda d: d.view(np.uint32).astype(np.uint64).sum(axis=1)
    # (a) the CPU recipe, seeded by Patchmatch::Initialize (matcher: 31x11 template, max_disp 128, cost 0.15)
    sp = O.seed_params(templ_cols=31, templ_rows=11, max_disp=128, max_matching_cost=0.15)
    seeds = O.cpu_initialize(l, r, 1, sp)
    sched = dict(noise_amp=[32.0, 8.0, 2.0, 0.5], patch_w=[5, 5, 3, 3], patch_h=[5, 5, 3, 3])
    prm = O.default_params(0, n_iters=4, bg_patch_w=3, bg_patch_h=3, bg_factor=1.5, left_right_check=0, nthreads=8,
                           literal=1, **sched)
    da, _ = O.match(prm, l, r, seeds, None)
    # (b) the GPU module's test: both views self-seeded by SparseInit(4)
    sl = O.sparse_init(l, r, 4, sp)
    sr = O.sparse_init(np.ascontiguousarray(r[:, ::-1]), np.ascontiguousarray(l[:, ::-1]), 4, sp)[:, ::-1]
    prm_b = O.default_params(1, n_iters=3, nthreads=8, cost_alpha=0.9)
    dbl, dbr = O.match(prm_b, l, r, sl, np.ascontiguousarray(sr))
    save("farmsim_fs1_376x240", left=l, right=r,
         cpu_recipe_rows=rowsum(da), cpu_recipe_total=np.uint64(rowsum(da).sum()), cpu_recipe_fg=np.float32((da > 0).mean()),
         gpu_test_rows_l=rowsum(dbl), gpu_test_rows_r=rowsum(dbr), gpu_test_fg=np.float32((dbl > 0).mean()))
    print("farmsim: seeds %.3f; CPU recipe foreground %.3f; GPU-test recipe foreground %.3f" %
          ((seeds > 0).mean(), (da > 0).mean(), (dbl > 0).mean()))


if __name__ == "__main__":
    if "--farmsim-only" in sys.argv:
        farmsim()
        sys.exit(0)
    if "--planes-only" in sys.argv:
        planes()
        sys.exit(0)
    synthetic()
    planes()
    if "--caddy" in sys.argv:
        caddy()
    if "--farmsim" in sys.argv:
        farmsim()

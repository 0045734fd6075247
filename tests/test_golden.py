"""Golden vectors committed under tests/golden/ (made by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them (-m "not gpu"), and the HIP engine must hit them
(-m gpu).  The reference ships no golden disparity maps (its stereo tests assert nothing), so these
pin the build against itself across rounds -- parity with the reference stays "unpinned"."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_same

CASES = ["synth64x48_cpu5", "synth64x48_gpu", "synth96x150_cpu_recipe"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def oracle_params(oracle, c):
    kw = dict(n_iters=int(c["iters"]), nthreads=8, left_right_check=int(c["lr"]))
    if "noise_amp" in c:
        kw.update(noise_amp=list(c["noise_amp"]), patch_w=list(c["patch_w"]), patch_h=list(c["patch_h"]),
                  bg_patch_w=int(c["bg_patch"]), bg_patch_h=int(c["bg_patch"]))
        return oracle.default_params(int(c["sem"]), **kw)
    return oracle.default_params(int(c["sem"]), patch=int(c["patch"]), **kw)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    c = load_case(name)
    dl, dr = oracle.match(oracle_params(oracle, c), c["left"], c["right"], c["seed_l"], c["seed_r"])
    assert_same(dl, c["disp_l"], "left")
    if int(c["lr"]):
        assert_same(dr, c["disp_r"], "right")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_reproduces_golden(pm, name):
    c = load_case(name)
    kw = dict(patchmatch_iters=int(c["iters"]), left_right_check=int(c["lr"]))
    if "noise_amp" in c:
        p = pm.default_params(int(c["sem"]), noise_amp=list(c["noise_amp"]), patch_w=list(c["patch_w"]),
                              patch_h=list(c["patch_h"]), bg_patch_w=int(c["bg_patch"]), bg_patch_h=int(c["bg_patch"]),
                              **kw)
    else:
        p = pm.default_params(int(c["sem"]), patch=int(c["patch"]), **kw)
    rows, cols = c["left"].shape
    with pm.Engine(p, max_rows=rows, max_cols=cols) as e:
        dl, dr = e.match(c["left"], c["right"], c["seed_l"], c["seed_r"])
    assert_same(dl, c["disp_l"], "left")
    if int(c["lr"]):
        assert_same(dr, c["disp_r"], "right")


def test_caddy_fixture_config1_plumbing(oracle):
    """BASELINE.json configs[0]: reference CPU PatchMatch on the 640x480 CADDY pair of the reference's
    test/resources, 3 iterations, 7x7 window -- checksum of the oracle output."""
    z = np.load(os.path.join(GOLDEN, "caddy_32_gray_640x480.npz"))
    l, r, sl, sr = z["left"], z["right"], z["seed_l"], z["seed_r"]
    assert l.shape == (480, 640) and l.dtype == np.uint8
    p = oracle.default_params(0, patch=7, n_iters=3, nthreads=8, left_right_check=0)
    dl, _ = oracle.match(p, l, r, sl, None)
    assert int(z["checksum_rows"].sum()) == int(np.asarray(z["checksum_total"]))
    got = (dl.view(np.uint32).astype(np.uint64).sum(axis=1))
    assert np.array_equal(got, z["checksum_rows"]), "per-row checksum of the 7x7 / 3-iteration disparity map"


@pytest.mark.gpu
def test_caddy_fixture_engine(pm):
    z = np.load(os.path.join(GOLDEN, "caddy_32_gray_640x480.npz"))
    p = pm.default_params(0, patch=7, patchmatch_iters=3, left_right_check=0)
    with pm.Engine(p, max_rows=480, max_cols=640) as e:
        dl, _ = e.match(z["left"], z["right"], z["seed_l"], None)
    got = (dl.view(np.uint32).astype(np.uint64).sum(axis=1))
    assert np.array_equal(got, z["checksum_rows"])


# ---- the reference's own PatchMatch test inputs -----------------------------------------------------------------
def _rowsum(d):
    return d.view(np.uint32).astype(np.uint64).sum(axis=1)


def test_farmsim_fixture_oracle(oracle):
    """fsl1.png / fsr1.png, the pair BOTH reference PatchMatch tests load (patchmatch_test.cpp:121-133,
    patchmatch_gpu_test.cpp:52-64), read as gray and halved with cv::resize to 376x240 (tests/golden/make_golden.py
    --farmsim).  The reference holds no expected output for them (parity stays unpinned); the fixture pins what the
    oracle computes for the two recipes, row by row."""
    z = np.load(os.path.join(GOLDEN, "farmsim_fs1_376x240.npz"))
    l, r = z["left"], z["right"]
    assert l.shape == (240, 376) and l.dtype == np.uint8
    sp = oracle.seed_params(templ_cols=31, templ_rows=11, max_disp=128, max_matching_cost=0.15)
    # (a) patchmatch_test.cpp:149-183
    seeds = oracle.cpu_initialize(l, r, 1, sp)
    sched = dict(noise_amp=[32.0, 8.0, 2.0, 0.5], patch_w=[5, 5, 3, 3], patch_h=[5, 5, 3, 3])
    prm = oracle.default_params(0, n_iters=4, bg_patch_w=3, bg_patch_h=3, bg_factor=1.5, left_right_check=0, nthreads=8,
                                literal=1, **sched)
    da, _ = oracle.match(prm, l, r, seeds, None)
    assert np.array_equal(_rowsum(da), z["cpu_recipe_rows"])
    assert int(z["cpu_recipe_rows"].sum()) == int(np.asarray(z["cpu_recipe_total"]))
    # the fused cost formula gives the same map as the literal patch + functor form
    prm.literal = 0
    db, _ = oracle.match(prm, l, r, seeds, None)
    assert np.array_equal(da, db)
    # (b) patchmatch_gpu_test.cpp:68-88
    sl = oracle.sparse_init(l, r, 4, sp)
    sr = np.ascontiguousarray(oracle.sparse_init(r[:, ::-1], l[:, ::-1], 4, sp)[:, ::-1])
    dl, dr = oracle.match(oracle.default_params(1, n_iters=3, nthreads=8, cost_alpha=0.9), l, r, sl, sr)
    assert np.array_equal(_rowsum(dl), z["gpu_test_rows_l"]) and np.array_equal(_rowsum(dr), z["gpu_test_rows_r"])
    assert (da > 0).mean() > 0.3 and (dl > 0).mean() > 0.2


@pytest.mark.gpu
def test_farmsim_fixture_engine(pm):
    """The HIP engine on the reference's own test pair, self-seeded exactly as the two tests run it: (a) the CPU
    recipe through Patchmatch::Initialize(il, ir, 1); (b) PatchmatchGpu::Match with cost_alpha 0.9 and 3 iterations,
    called five times in a row on the same handle."""
    z = np.load(os.path.join(GOLDEN, "farmsim_fs1_376x240.npz"))
    l, r = z["left"], z["right"]
    rows, cols = l.shape
    sched = dict(noise_amp=[32.0, 8.0, 2.0, 0.5], patch_w=[5, 5, 3, 3], patch_h=[5, 5, 3, 3])
    pa = pm.default_params(0, patchmatch_iters=4, bg_patch_w=3, bg_patch_h=3, win_by_factor=1.5, left_right_check=0,
                           sparse_init=1, cpu_initialize_factor=1, templ_cols=31, templ_rows=11, max_disp=128,
                           max_matching_cost=0.15, **sched)
    with pm.Engine(pa, max_rows=rows, max_cols=cols) as e:
        da, _ = e.match(l, r)
    assert np.array_equal(_rowsum(da), z["cpu_recipe_rows"]), "patchmatch_test.cpp:149-183 on fsl1 / fsr1"
    pb = pm.default_params(1, patchmatch_iters=3, cost_alpha=0.9, sparse_init=1, templ_cols=31, templ_rows=11,
                           max_disp=128, max_matching_cost=0.15)
    with pm.Engine(pb, max_rows=rows, max_cols=cols) as e:
        for i in range(5):
            dl, dr = e.match(l, r)
            assert np.array_equal(_rowsum(dl), z["gpu_test_rows_l"]), f"patchmatch_gpu_test.cpp:68-88, call {i} (left)"
            assert np.array_equal(_rowsum(dr), z["gpu_test_rows_r"]), f"call {i} (right)"

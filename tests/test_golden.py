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

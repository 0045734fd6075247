"""The C oracle's algorithm layer: hand-derivable cases, agreement with the independent pure-Python
restatement (small images), literal-vs-fused cost, thread-count independence and the properties the
domain guarantees."""
import numpy as np
import pytest

import pyref
from conftest import assert_same, small_pair


def tiny(synth, idx=3, rows=20, cols=28):
    l, r, sl, sr, gt = small_pair(synth, idx, rows, cols, n_points=12, dilate_factor=1)
    return l, r, sl, sr


# ---- cost functor -------------------------------------------------------------------------------
def test_functor_known_answers(oracle):
    pl = np.full(9, 100, np.uint8)
    pr = np.full(9, 90, np.uint8)
    g0 = np.zeros(9, np.float32)
    # colour mean 10, gradient 0 -> 0.7*10
    assert oracle.cpu_functor(pl, pr, g0, g0) == np.float32(np.float32(0.7) * np.float32(10.0))
    # truncation at tau_color = 50 and tau_grad = 20
    pr = np.zeros(9, np.uint8)
    g1 = np.full(9, 40.0, np.float32)
    exp = np.float32(np.float32(0.7) * np.float32(50.0)) + np.float32((np.float32(1) - np.float32(0.7)) * np.float32(20.0))
    assert oracle.cpu_functor(pl, pr, g0, g1) == np.float32(exp)
    # gradients saturate at 255 before differencing (Image1f -> Image1b): 300 vs 1000 -> 0
    assert oracle.cpu_functor(pl, pl, np.full(9, 300.0, np.float32), np.full(9, 1000.0, np.float32)) == 0.0
    # round half to even: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2
    a = np.array([0.5, 1.5, 2.5] * 3, np.float32)
    b = np.zeros(9, np.float32)
    assert oracle.cpu_functor(pl, pl, a, b) == np.float32(np.float32(0.3) * np.float32(np.float64(12) * (1.0 / 9)))\
        or oracle.cpu_functor(pl, pl, a, b) == np.float32((np.float32(1) - np.float32(0.7)) * np.float32(np.float64(12) * (1.0 / 9)))


def test_cost_literal_equals_direct_and_pyref(oracle, synth):
    l, r, _, _ = tiny(synth, 4, 24, 40)
    ims = oracle.ImageSet(l, r)
    rng = np.random.default_rng(2)
    for pw, ph in ((3, 3), (5, 5), (7, 3), (11, 11)):
        for _ in range(60):
            x = int(rng.integers(pw // 2, 40 - pw // 2))
            y = int(rng.integers(ph // 2, 24 - ph // 2))
            d = float(np.float32(rng.uniform(0, x - pw // 2))) if x > pw // 2 else 0.0
            if rng.random() < 0.2:
                d = float(int(d))
            a = oracle.cpu_cost(ims, pw, ph, x, y, d, literal=True)
            b = oracle.cpu_cost(ims, pw, ph, x, y, d, literal=False)
            c = float(pyref.cpu_cost(ims.il, ims.ir, ims.gl, ims.gr, pw, ph, x, y, d))
            assert a == b == c, (pw, ph, x, y, d, a, b, c)


def test_gpu_cost5_matches_pyref(oracle, synth):
    l, r, _, _ = tiny(synth, 5, 16, 30)
    ims = oracle.ImageSet(l, r)
    rng = np.random.default_rng(9)
    for _ in range(200):
        x, y = int(rng.integers(1, 29)), int(rng.integers(1, 15))
        xr = float(np.float32(max(x - rng.uniform(0, x), 1.0)))
        a = oracle.gpu_cost5(ims, y, x, float(y), xr)
        b = float(pyref.gpu_cost5(ims.il, ims.ir, ims.gl, ims.gr, y, x, y, xr))
        assert a == b


# ---- noise --------------------------------------------------------------------------------------
def test_add_noise(oracle):
    d = np.zeros((6, 8), np.float32)
    d[2:4, 3:6] = 10.0
    out = oracle.cpu_add_noise(d, 2.0, (d > 0).astype(np.uint8) * 255)
    assert np.all(out[d == 0] == 0)
    noise = oracle.rng_fill_uniform(48, -2.0, 2.0).reshape(6, 8)
    assert np.array_equal(out[d > 0], (d + noise)[d > 0])
    assert_same(out, pyref.cpu_add_noise(d, 2.0, d > 0), "pyref")
    # no mask: every pixel, clamped at 0
    out = oracle.cpu_add_noise(d, 2.0, None)
    assert np.array_equal(out, np.maximum(d + noise, 0))
    # the two reference forms agree (Q2): GPU form on unit noise == CPU form
    unit = oracle.rng_fill_uniform(48, -1.0, 1.0).reshape(6, 8)
    for amp in (32.0, 8.0, 0.5):
        a = oracle.gpu_add_foreground_noise(d, unit, amp)
        b = oracle.cpu_add_noise(d, amp, (d > 0).astype(np.uint8))
        assert_same(a, b, "gpu vs cpu noise form")
        assert_same(a, pyref.gpu_add_foreground_noise(d, unit, amp), "pyref gpu noise")


# ---- sweeps -------------------------------------------------------------------------------------
def test_propagate_constant_images_only_clamps(oracle):
    # all costs are 0 -> no strict improvement anywhere -> interior becomes clamp(d, 0, x - pw/2)
    l = np.full((12, 16), 77, np.uint8)
    ims = oracle.ImageSet(l, l)
    rng = np.random.default_rng(1)
    d = (rng.random((12, 16)) * 20).astype(np.float32)
    out = oracle.cpu_propagate(ims, d, 3, 3)
    exp = d.copy()
    xs = np.arange(16, dtype=np.float32)[None, :]
    exp[1:-1, 1:-1] = np.minimum(d, xs - 1)[1:-1, 1:-1]
    assert_same(out, exp)


def test_propagate_adopts_true_disparity_along_a_row(oracle):
    # right = texture, left = right shifted by exactly 4 px: d = 4 has cost 0.  One good pixel at the
    # start of every row must spread over the row in pass A (left neighbour), nothing else changes it.
    rng = np.random.default_rng(0)
    r = rng.integers(0, 256, (9, 40), dtype=np.uint8)
    l = np.roll(r, 4, axis=1)
    ims = oracle.ImageSet(l, r)
    d = np.full((9, 40), 9.0, np.float32)
    d[:, 6] = 4.0
    out = oracle.cpu_propagate(ims, d, 3, 3, pass_mask=1)
    assert np.all(out[1:-1, 6:-1] == 4.0)
    # left of the good pixel: clamp(9, 0, x-1) or an adopted (equally clamped) neighbour value
    assert np.all(out[1:-1, 1:6] <= np.arange(1, 6) - 1.0) and np.all(out[1:-1, 5] == 4.0)
    assert np.all(out[0] == d[0]) and np.all(out[-1] == d[-1]) and np.all(out[:, 0] == 9.0) and np.all(out[:, -1] == 9.0)
    # reverse pass spreads it to the left as far as the candidate stays valid (x - 4 >= 1)
    out2 = oracle.cpu_propagate(ims, out, 3, 3, pass_mask=4)
    assert np.all(out2[1:-1, 5:-1] == 4.0)


@pytest.mark.parametrize("pw,ph", [(3, 3), (5, 3), (5, 5)])
def test_propagate_matches_pyref(oracle, synth, pw, ph):
    l, r, sl, _ = tiny(synth, 6, 14, 22)
    ims = oracle.ImageSet(l, r)
    d = oracle.cpu_add_noise(sl, 4.0, (sl > 0).astype(np.uint8))
    for mask in (1, 2, 4, 8, 15):
        a = oracle.cpu_propagate(ims, d, ph, pw, pass_mask=mask)
        b = pyref.cpu_propagate(ims.il, ims.ir, ims.gl, ims.gr, d, ph, pw, pass_mask=mask)
        assert_same(a, b, f"pass_mask {mask}")
    a = oracle.cpu_remove_background(ims, d, ph, pw, 1.5)
    assert_same(a, pyref.cpu_remove_background(ims.il, ims.ir, ims.gl, ims.gr, d, ph, pw, 1.5), "background")


def test_gpu_semantics_matches_pyref(oracle, synth):
    l, r, sl, sr = tiny(synth, 8, 12, 20)
    ims = oracle.ImageSet(l, r)
    unit = oracle.rng_fill_uniform(12 * 20, -1.0, 1.0).reshape(12, 20)
    d = oracle.gpu_add_foreground_noise(sl, unit, 4.0)
    cur = d
    for k, (axis, direction) in enumerate(((0, 1), (1, 1), (0, -1), (1, -1))):
        a = oracle.gpu_propagate(ims, cur, pass_mask=1 << k)
        b = pyref.gpu_propagate(ims.il, ims.ir, ims.gl, ims.gr, cur, axis, direction)
        assert_same(a, b, f"sweep {k}")
        cur = a
    assert_same(oracle.gpu_mask_background(ims, cur), pyref.gpu_mask_background(ims.il, ims.ir, ims.gl, ims.gr, cur),
                "mask background")
    assert_same(oracle.gpu_mask_occlusions(cur, sr), pyref.gpu_mask_occlusions(cur, sr), "mask occlusions")


def test_mask_occlusions_known_answers(oracle):
    dl = np.array([[0, 2, 4, 4]], np.float32)
    dr = np.array([[0, 5, 4, 9]], np.float32)
    # x=1: dr[(int)max(1-2,0)=0]=0 < 0.7*2 -> 0;  x=2: dr[0]=0 < 2.8 -> 0;  x=3: dr[0] -> 0; x=0: dl=0, dr[0]=0 keeps 0
    assert np.array_equal(oracle.gpu_mask_occlusions(dl, dr), [[0, 0, 0, 0]])
    dl = np.array([[0, 0, 0, 3, 3]], np.float32)
    dr = np.array([[3, 4.3, 2.0, 0, 0]], np.float32)
    # x=3 -> dr[0]=3 in [2.1,4.2] keep;  x=4 -> dr[1]=4.3 > 4.2 -> 0
    assert np.array_equal(oracle.gpu_mask_occlusions(dl, dr), [[0, 0, 0, 3, 0]])


# ---- pipelines ----------------------------------------------------------------------------------
@pytest.mark.parametrize("sem", [0, 1])
def test_match_thread_count_and_literal_do_not_change_results(oracle, synth, sem):
    l, r, sl, sr, _ = small_pair(synth, 2, 40, 64, n_points=24, dilate_factor=2)
    base = oracle.match(oracle.default_params(sem, patch=5, n_iters=2, nthreads=1), l, r, sl, sr)
    for kw in ({"nthreads": 8}, {"nthreads": 3, "literal": 1}):
        other = oracle.match(oracle.default_params(sem, patch=5, n_iters=2, **kw), l, r, sl, sr)
        assert_same(base[0], other[0], f"left {kw}")
        assert_same(base[1], other[1], f"right {kw}")


@pytest.mark.parametrize("sem", [0, 1])
def test_match_properties(oracle, synth, sem):
    rows, cols = 48, 80
    l, r, sl, sr, gt = small_pair(synth, 9, rows, cols, n_points=30, dilate_factor=2)
    prm = oracle.default_params(sem, patch=5, n_iters=3, nthreads=8)
    dl, dr = oracle.match(prm, l, r, sl, sr)
    assert dl.min() >= 0 and dr.min() >= 0 and np.isfinite(dl).all() and np.isfinite(dr).all()
    half = 2 if sem == 0 else 1
    xs = np.arange(cols, dtype=np.float32)[None, :]
    inner = (slice(half, rows - half), slice(half, cols - half))
    if sem == 0:
        # patchmatch.cpp:175 clamps every visited pixel: d <= x - pw/2 (the CUDA rule only clamps
        # what it adopts, patchmatch_gpu.cu:169, so PM_SEM_GPU has no such guarantee)
        assert np.all(dl[inner] <= (xs - half)[:, half:cols - half])
        # right view (mirrored): d <= (W-1-x) - pw/2
        assert np.all(dr[inner] <= (cols - 1 - xs - half)[:, half:cols - half])
    # no seeds -> everything stays background
    zl, zr = oracle.match(prm, l, r, None, None)
    assert not zl.any() and not zr.any()
    # deterministic
    dl2, dr2 = oracle.match(prm, l, r, sl, sr)
    assert_same(dl, dl2)
    assert_same(dr, dr2)
    # mirrored problem: matching (flip R, flip L) gives the flipped maps with the views swapped
    ml, mr = oracle.match(prm, r[:, ::-1], l[:, ::-1], sr[:, ::-1], sl[:, ::-1])
    prm_nolr = oracle.default_params(sem, patch=5, n_iters=3, nthreads=8, left_right_check=0)
    ul, _ = oracle.match(prm_nolr, l, r, sl, None)
    vl, _ = oracle.match(prm_nolr, r[:, ::-1], l[:, ::-1], sr[:, ::-1], None)
    assert_same(vl[:, ::-1], dr, "right view == left view of the mirrored pair")
    assert (np.abs(ul - gt)[ul > 0] < 1.0).mean() > 0.9

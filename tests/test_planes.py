"""PM_MODE_PLANES: slanted-plane PatchMatch (random init, red-black spatial propagation, view propagation, random
plane refinement, windowed cost) -- the kernels BASELINE.json's north_star names.

The reference has no such code (src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:379-411 keeps a scalar disparity), so
the checker is this build's own CPU definition, oracle/pm_planes_oracle.c.  Bars:
  * HIP == definition bit for bit on the same seeded random numbers, stage by stage and end to end, for f32 AND f16
    state (the f16 rounding points are part of the definition, so f16 is exact too);
  * quality against the synthetic ground truth, with the tolerance stated in the test: f32 planes >= 97 % of the
    valid pixels within 1 px; f16 state may lose at most 0.5 percentage points against f32.
"""
import numpy as np
import pytest

from conftest import assert_same

ROWS, COLS = 72, 200  # not multiples of the 8 x 32 / 8 x 64 tiles


def okw(pm_params):
    """pm_params (ctypes) -> keyword arguments of the oracle's parameter struct."""
    import oracle_lib
    return oracle_lib.planes_kwargs_of(pm_params, nthreads=8)


def pparams(pm, patch=11, iters=2, f16=0, **kw):
    kw.setdefault("state_dtype", f16)
    kw.setdefault("mode", pm.PM_MODE_PLANES)
    return pm.default_params(0, patch=patch, patchmatch_iters=iters, **kw)


# ---- the definition itself (CPU) ---------------------------------------------------------------------------------
def test_rand_is_a_pure_function_of_its_key(oracle):
    a = [oracle.planes_rand(123, 1, 2, 1, 0, d, 17, 5) for d in range(3)]
    assert a == [oracle.planes_rand(123, 1, 2, 1, 0, d, 17, 5) for d in range(3)]
    assert len({oracle.planes_rand(123, 1, 2, 1, 0, 0, x, 5) for x in range(256)}) == 256
    # python restatement of the key mix + one cv::RNG multiply-with-carry step
    M = (1 << 64) - 1

    def ref(seed, stage, it, k, view, draw, x, y):
        tag = stage | (it << 4) | (k << 12) | (view << 20) | (draw << 24)
        s = (seed + 0x9E3779B97F4A7C15 * (tag + 1)) & M
        s ^= (y << 32) | x
        s ^= s >> 30
        s = (s * 0xBF58476D1CE4E5B9) & M
        s ^= s >> 27
        s = (s * 0x94D049BB133111EB) & M
        s ^= s >> 31
        s = ((s & 0xffffffff) * 4164903690 + (s >> 32)) & M
        return s & 0xffffffff

    for key in [(123, 0, 0, 0, 0, 0, 0, 0), (123, 1, 7, 2, 1, 2, 1279, 719), (99, 1, 3, 0, 1, 1, 40, 41)]:
        assert oracle.planes_rand(*key) == ref(*key)
    # roughly uniform
    v = np.array([oracle.planes_rand(123, 0, 0, 0, 0, 0, x, y) for y in range(64) for x in range(64)], np.float64)
    assert abs(v.mean() / 2 ** 32 - 0.5) < 0.02


def test_quant_f16_equals_numpy_float16(oracle):
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.normal(0, 1, 2000), rng.uniform(-70000, 70000, 2000), rng.uniform(-1e-4, 1e-4, 2000),
        rng.uniform(-7e-8, 7e-8, 500), np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25,
                                                   2.0 ** -25 * 1.0001, 2.0 ** -14, 1.0 + 2.0 ** -11,
                                                   1.0 + 3 * 2.0 ** -11, 128.0625, 127.96875])]).astype(np.float32)
    # every f16 midpoint in a few binades (ties to even)
    h = np.arange(0x3c00, 0x3c40, dtype=np.uint16).view(np.float16).astype(np.float32)
    vals = np.concatenate([vals, (h[:-1] + h[1:]) / 2]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).astype(np.float32)
    got = np.array([oracle.planes_quant_f16(v) for v in vals], np.float32)
    assert np.array_equal(got, want)


def test_cost_of_the_true_plane_is_zero_on_a_shifted_texture(oracle):
    rng = np.random.default_rng(2)
    right = rng.integers(0, 256, (40, 120), dtype=np.uint8)
    left = np.roll(right, 7, axis=1)  # left(x) = right(x - 7)
    v = oracle.PlanesViews(left, right)
    p = oracle.planes_params(patch=11)
    assert v.cost(p, 0, 60, 20, 0.0, 0.0, 7.0) == 0.0
    assert v.cost(p, 0, 60, 20, 0.0, 0.0, 9.0) > 5.0
    # hand value for a 3x3 window of constant images: |10 - 30| colour, zero gradient
    l2 = np.full((8, 16), 10, np.uint8)
    r2 = np.full((8, 16), 30, np.uint8)
    v2 = oracle.PlanesViews(l2, r2)
    p3 = oracle.planes_params(patch=3)
    assert v2.cost(p3, 0, 8, 4, 0.1, -0.2, 1.5) == np.float32(np.float32(0.7) * np.float32(np.float32(180) * np.float32(1.0 / 9)))


def test_definition_recovers_the_synthetic_truth_and_f16_costs_little(oracle, synth):
    p = synth.make_pair(3, 120, 240)
    res = {}
    for f16 in (0, 1):
        prm = oracle.planes_params(n_iters=6, nthreads=8, state_f16=f16)
        dl, dr = oracle.planes_match(prm, p["left"], p["right"])
        ok = dl > 0
        err = np.abs(dl - p["gt"])
        res[f16] = ((err[ok] < 1).mean(), ok.mean())
        assert res[f16][1] > 0.5
    assert res[0][0] >= 0.97, res                  # stated tolerance: >= 97 % of valid pixels within 1 px
    assert res[1][0] >= res[0][0] - 0.005, res     # f16 state: at most 0.5 points worse
    # thread count does not matter
    prm1 = oracle.planes_params(n_iters=2, nthreads=1)
    prm8 = oracle.planes_params(n_iters=2, nthreads=8)
    a = oracle.planes_match(prm1, p["left"][:48, :96], p["right"][:48, :96])
    b = oracle.planes_match(prm8, p["left"][:48, :96], p["right"][:48, :96])
    assert_same(a[0], b[0], "threads")


def test_two_neighbour_option_of_the_spatial_stage(oracle, synth):
    """PMO_PL_NEIGH_TWO: the colour passes of an even iteration offer the left and the upper neighbour only, those of an odd
    iteration the right and the lower one.  A zero-cost plane planted in ONE pixel of an otherwise bad field travels
    accordingly; the whole Match keeps the quality bar of the definition (>= 97 % of the valid pixels within 1 px here)."""
    rng = np.random.default_rng(5)
    right = rng.integers(0, 256, (40, 120), dtype=np.uint8)
    left = np.roll(right, 7, axis=1)  # the true plane everywhere: (0, 0, 7)
    for neighbours in (oracle.PL_NEIGH_FOUR, oracle.PL_NEIGH_TWO):
        for it in (0, 1):
            v = oracle.PlanesViews(left, right)
            prm = oracle.planes_params(patch=5, neighbours=neighbours, nthreads=1)
            a, b, z, c = v.planes[0]
            a[:], b[:] = 0.0, 0.0
            z[:] = 3.0                        # a wrong plane everywhere ...
            for y in range(40):
                for x in range(120):
                    c[y, x] = v.cost(prm, 0, x, y, 0.0, 0.0, 3.0)
            y0, x0 = 20, 60                   # ... except one pixel (x0 + y0 even: colour 0)
            z[y0, x0] = 7.0
            c[y0, x0] = v.cost(prm, 0, x0, y0, 0.0, 0.0, 7.0)
            assert c[y0, x0] == 0.0
            v.spatial(prm, 0, 1 + 2 * it)     # the pass of the OTHER colour: the four neighbours of (x0, y0) may adopt
            got = {(dx, dy): float(z[y0 + dy, x0 + dx]) == 7.0 for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1))}
            if neighbours == oracle.PL_NEIGH_FOUR:
                assert all(got.values()), (neighbours, it, got)
            elif it == 0:  # left + up are offered: the pixels to the RIGHT of and BELOW the planted one see it
                assert got == {(1, 0): True, (-1, 0): False, (0, 1): True, (0, -1): False}, got
            else:
                assert got == {(1, 0): False, (-1, 0): True, (0, 1): False, (0, -1): True}, got
    p = synth.make_pair(3, 120, 240)
    prm = oracle.planes_params(n_iters=6, nthreads=8, neighbours=oracle.PL_NEIGH_TWO)
    dl, _ = oracle.planes_match(prm, p["left"], p["right"])
    ok = dl > 0
    assert ok.mean() > 0.5 and (np.abs(dl - p["gt"])[ok] < 1).mean() >= 0.97


def test_golden_planes_fixture(oracle):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "planes_64x96.npz"))
    for f16 in (0, 1):
        for window, tag in ((oracle.PL_WINDOW_CHECKER, ""), (oracle.PL_WINDOW_FULL, "full_")):
            prm = oracle.planes_params(n_iters=3, nthreads=4, state_f16=f16, patch=7, max_disp=24, window=window)
            dl, dr = oracle.planes_match(prm, g["left"], g["right"])
            assert_same(dl, g[f"{tag}disp_l_f{16 if f16 else 32}"], f"golden left {tag}")
            assert_same(dr, g[f"{tag}disp_r_f{16 if f16 else 32}"], f"golden right {tag}")


# ---- HIP vs definition --------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def dev_pair(pair_list):
    import torch
    dev = torch.device("cuda:0")
    t = lambda k, dt: torch.from_numpy(np.stack([p[k] for p in pair_list])).to(dev).to(dt).contiguous()
    return t("left", torch.uint8), t("right", torch.uint8), t("seed_l", torch.float32), t("seed_r", torch.float32)


@gpu
@pytest.mark.parametrize("window,neighbours", [(1, 0), (0, 0), (1, 1)])  # (1, 0) = the defaults
@pytest.mark.parametrize("f16", [0, 1])
@pytest.mark.parametrize("seeded", [False, True])
def test_each_stage_matches_the_definition(pm, oracle, synth, f16, seeded, window, neighbours):
    """NS-1 random plane initialisation, NS-2 red-black propagation, NS-3 view propagation, NS-4 refinement -- on the
    checkerboard window (the mode's default) and on the full one, and with the spatial stage's two-neighbour option
    (left + up in the passes of an even iteration, right + down in those of an odd one)."""
    import torch
    p = synth.make_pair(11, ROWS, COLS, n_points=30, dilate_factor=2)
    prm = pparams(pm, iters=2, f16=f16, max_disp=48, plane_window=window, plane_neighbours=neighbours)
    op = oracle.planes_params(**okw(prm))
    L, R, SL, SR = dev_pair([p])
    ov = oracle.PlanesViews(p["left"], p["right"])
    names = "a b z cost".split()

    def check(e, what):
        torch.cuda.synchronize()
        for v in range(2):
            got = e.planes_read(0, v)
            for k in range(4):
                assert_same(got[k], ov.planes[v][k], f"{what}: view {v} {names[k]}")

    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
        e.planes_begin(1, L.data_ptr(), R.data_ptr(), ROWS, COLS, SL.data_ptr() if seeded else None,
                       SR.data_ptr() if seeded else None)
        for v in range(2):
            seed = None
            if seeded:
                seed = p["seed_l"] if v == 0 else p["seed_r"][:, ::-1]
            ov.init(op, v, seed)
        check(e, "init")
        for it in range(2):
            for par in (0, 1):
                e.planes_step(pm.PM_PL_SPATIAL, par + 2 * it)
                for v in range(2):
                    ov.spatial(op, v, par + 2 * it)
                check(e, f"spatial it {it} colour {par}")
            if it == 0:  # the two stages on their own ...
                for v in range(2):
                    e.planes_step(pm.PM_PL_VIEW, v)
                    ov.view_prop(op, v)
                    check(e, f"view propagation it {it} into view {v}")
                e.planes_step(pm.PM_PL_REFINE, it)
                for v in range(2):
                    ov.refine(op, v, it)
                check(e, f"refine it {it}")
            else:        # ... and fused as Match() runs them: per view, propagation then refinement in one launch
                for v in range(2):
                    e.planes_step(pm.PM_PL_VIEW_REFINE, 2 * it + v)
                    ov.view_prop(op, v)
                    ov.refine(op, v, it)
                    check(e, f"fused view propagation + refinement it {it} view {v}")


@gpu
def test_stages_on_injected_adversarial_planes(pm, oracle, synth):
    """Planes at the bounds (slopes +-slope_max, z = 0 / max_disp / beyond the column), steep planes next to flat
    ones, equal neighbours: written into the engine and into the definition, then every stage once."""
    import torch
    p = synth.make_pair(12, ROWS, COLS)
    prm = pparams(pm, iters=1, max_disp=40)
    op = oracle.planes_params(**okw(prm))
    L, R, _, _ = dev_pair([p])
    ov = oracle.PlanesViews(p["left"], p["right"])
    rng = np.random.default_rng(5)
    ys, xs = np.mgrid[0:ROWS, 0:COLS]
    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
        e.planes_begin(1, L.data_ptr(), R.data_ptr(), ROWS, COLS)
        for v in range(2):
            a = rng.choice(np.array([-1.0, 1.0, 0.0, 0.999, -0.5], np.float32), (ROWS, COLS))
            b = rng.choice(np.array([-1.0, 1.0, 0.0, 0.25], np.float32), (ROWS, COLS))
            z = rng.choice(np.array([0.0, 40.0, 39.99, 12.5, 3.0], np.float32), (ROWS, COLS))
            z = np.minimum(z, xs.astype(np.float32))  # keep the state admissible (0 <= z <= min(max_disp, x))
            a[:, 50:60] = 0.25
            b[:, 50:60] = 0.0
            z[:, 50:60] = 10.0  # a block of identical planes (equal candidates are never evaluated)
            ov.planes[v][0], ov.planes[v][1], ov.planes[v][2] = a, b, z
            for yy in range(ROWS):
                for xx in range(COLS):
                    ov.planes[v][3][yy, xx] = ov.cost(op, v, xx, yy, a[yy, xx], b[yy, xx], z[yy, xx])
            e.planes_write(0, v, ov.planes[v])
        for stage, args in ((pm.PM_PL_SPATIAL, (0, 1)), (pm.PM_PL_VIEW, (0, 1)), (pm.PM_PL_REFINE, (0,))):
            for arg in args:
                e.planes_step(stage, arg)
                if stage == pm.PM_PL_SPATIAL:
                    for v in range(2):
                        ov.spatial(op, v, arg)
                elif stage == pm.PM_PL_VIEW:
                    ov.view_prop(op, arg)
                else:
                    for v in range(2):
                        ov.refine(op, v, arg)
                torch.cuda.synchronize()
                for v in range(2):
                    got = e.planes_read(0, v)
                    for k in range(4):
                        assert_same(got[k], ov.planes[v][k], f"stage {stage} arg {arg} view {v} plane {k}")


@gpu
@pytest.mark.parametrize("window,neighbours", [(1, 0), (0, 0), (1, 1)])  # checkerboard + four neighbours = the defaults
@pytest.mark.parametrize("f16", [0, 1])
@pytest.mark.parametrize("patch,max_disp", [(11, 64), (7, 32), (5, 128), (3, 16), (15, 40), (11, 300)])  # 300: > 64 KB of LDS
def test_match_equals_the_definition(pm, oracle, synth, f16, patch, max_disp, window, neighbours):
    p = synth.make_pair(20 + patch, ROWS, COLS)
    prm = pparams(pm, patch=patch, iters=3, f16=f16, max_disp=max_disp, plane_window=window, plane_neighbours=neighbours)
    want = oracle.planes_match(oracle.planes_params(**okw(prm)), p["left"], p["right"])
    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
        got = e.match(p["left"], p["right"])
    assert_same(got[0], want[0], "left map")
    assert_same(got[1], want[1], "right map")


@gpu
def test_match_variants(pm, oracle, synth):
    """seed maps, one view only, other schedule / slopes / seed, batch == singles, pipelined entry point."""
    pairs = [synth.make_pair(30 + i, ROWS, COLS, n_points=40, dilate_factor=2) for i in range(3)]
    p = pairs[0]
    # seeded
    prm = pparams(pm, iters=2, max_disp=48)
    want = oracle.planes_match(oracle.planes_params(**okw(prm)), p["left"], p["right"], p["seed_l"], p["seed_r"])
    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
        got = e.match(p["left"], p["right"], p["seed_l"], p["seed_r"])
    assert_same(got[0], want[0], "seeded left")
    assert_same(got[1], want[1], "seeded right")
    # one view, no consistency mask
    prm = pparams(pm, iters=2, max_disp=48, left_right_check=0)
    want = oracle.planes_match(oracle.planes_params(**okw(prm)), p["left"], p["right"])
    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
        got = e.match(p["left"], p["right"])
    assert_same(got[0], want[0], "one view")
    # other constants
    prm = pparams(pm, iters=2, max_disp=48, plane_refine_steps=5, plane_slope_max=0.5, plane_slope_init=0.5,
                  plane_slope_per_disp=0.03, plane_lr_tol=0.5, noise_seed=77, noise_amp=[8, 3, 1] + [0.5] * 13,
                  functor_alpha=0.5, functor_tau_color=30.0, functor_tau_grad=40.0)
    want = oracle.planes_match(oracle.planes_params(**okw(prm)), p["left"], p["right"])
    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS) as e:
        got = e.match(p["left"], p["right"])
    assert_same(got[0], want[0], "other constants left")
    assert_same(got[1], want[1], "other constants right")
    # batch of 3 == 3 singles == pipelined
    prm = pparams(pm, iters=2, max_disp=48)
    op = oracle.planes_params(**okw(prm))
    with pm.Engine(prm, max_rows=ROWS, max_cols=COLS, max_batch=3) as e:
        dls, drs = e.match_batch([q["left"] for q in pairs], [q["right"] for q in pairs])
        for i, q in enumerate(pairs):
            e.submit(q["left"], q["right"], tag=i)
        piped = [e.collect() for _ in pairs]
    for i, q in enumerate(pairs):
        want = oracle.planes_match(op, q["left"], q["right"])
        assert_same(dls[i], want[0], f"batch left {i}")
        assert_same(drs[i], want[1], f"batch right {i}")
        assert_same(piped[i][0], want[0], f"pipelined left {i}")
        assert piped[i][2] == i


@gpu
def test_self_seeded_planes_match(pm, oracle, synth):
    """sparse_init = 1: the device SparseInit (patchmatch_gpu.cu:414-442) feeds the plane initialisation."""
    p = synth.make_pair(41, 96, 256)
    prm = pparams(pm, iters=2, max_disp=64, sparse_init=1, init_dilate_factor=2)
    sp = oracle.seed_params(max_disp=64)  # the seeder shares max_disp with the plane search
    sl = oracle.sparse_init(p["left"], p["right"], 2, sp)
    sr_m = oracle.sparse_init(p["right"][:, ::-1], p["left"][:, ::-1], 2, sp)
    want = oracle.planes_match(oracle.planes_params(**okw(prm)), p["left"], p["right"], sl,
                               np.ascontiguousarray(sr_m[:, ::-1]))
    with pm.Engine(prm, max_rows=96, max_cols=256) as e:
        got = e.match(p["left"], p["right"])
    assert_same(got[0], want[0], "self-seeded left")
    assert_same(got[1], want[1], "self-seeded right")


@gpu
def test_golden_planes_fixture_on_device(pm):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "planes_64x96.npz"))
    for f16 in (0, 1):
        for window, tag in ((pm.PM_PL_WINDOW_CHECKER, ""), (pm.PM_PL_WINDOW_FULL, "full_")):
            prm = pparams(pm, patch=7, iters=3, f16=f16, max_disp=24, plane_window=window)
            with pm.Engine(prm, max_rows=64, max_cols=96) as e:
                dl, dr = e.match(g["left"], g["right"])
            assert_same(dl, g[f"{tag}disp_l_f{16 if f16 else 32}"], f"golden left {tag}")
            assert_same(dr, g[f"{tag}disp_r_f{16 if f16 else 32}"], f"golden right {tag}")


@gpu
def test_parameter_errors(pm):
    for kw in (dict(patch=4), dict(plane_slope_max=0.0), dict(state_dtype=2), dict(max_disp=0), dict(mode=2),
               dict(plane_refine_steps=-1), dict(patch=15, max_disp=1024)):  # the last one: the tile exceeds the LDS
        with pytest.raises(pm.PmError) as ei:
            pm.Engine(pparams(pm, **kw), max_rows=64, max_cols=64)
        assert ei.value.status == pm.PM_ERR_INVALID_ARG
    # stage entry points refuse a scalar-mode handle, and a planes handle before pm_planes_begin
    with pm.Engine(pm.default_params(0), max_rows=64, max_cols=64) as e:
        with pytest.raises(pm.PmError):
            e.planes_step(pm.PM_PL_SPATIAL, 0)
    with pm.Engine(pparams(pm), max_rows=64, max_cols=64) as e:
        with pytest.raises(pm.PmError):
            e.planes_step(pm.PM_PL_SPATIAL, 0)


@gpu
@pytest.mark.parametrize("f16", [0, 1])
def test_full_size_properties(pm, oracle, synth, f16):
    """1280x720, 8 iterations, 11x11 (BASELINE configs[1] / configs[4] shapes): run-to-run determinism, quality
    against the synthetic truth (>= 97 % of valid pixels within 1 px; f16 within 0.5 points of f32 is checked at
    small size against the definition), a 40-row band of pixels re-costed by the definition, and the consistency
    mask's own invariant."""
    import torch
    rows, cols = 720, 1280
    p = synth.make_pair(0, rows, cols)
    prm = pparams(pm, iters=8, f16=f16)
    op = oracle.planes_params(**okw(prm))
    L, R, _, _ = dev_pair([p])
    dev = L.device
    DL = torch.empty((1, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        e.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
        e.synchronize()
        first = DL.clone()
        e.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
        e.synchronize()
        assert torch.equal(first, DL)
        planes = [e.planes_read(0, v) for v in range(2)]
    dl, dr = DL[0].cpu().numpy(), DR[0].cpu().numpy()
    ok = dl > 0
    err = np.abs(dl - p["gt"])
    assert ok.mean() > 0.8
    assert (err[ok] < 1).mean() >= 0.97
    # state invariants: admissible planes, stored cost == cost of the stored plane under the definition
    xs = np.arange(cols, dtype=np.float32)[None, :]
    ov = oracle.PlanesViews(p["left"], p["right"])
    for v in range(2):
        a, b, z, c = planes[v]
        assert (np.abs(a) <= 1).all() and (np.abs(b) <= 1).all()
        assert (z >= 0).all() and (z <= np.minimum(128.0, xs)).all()
        for y in range(340, 380, 13):
            for x in range(7, cols, 97):
                want = ov.cost(op, v, x, y, a[y, x], b[y, x], z[y, x])
                if f16:
                    want = oracle.planes_quant_f16(want)
                assert c[y, x] == np.float32(want), (v, x, y)
    # output maps are the z planes; masked pixels are exactly those failing |dl - dr(x - dl)| <= tol
    assert_same(dr, planes[1][2][:, ::-1], "right map = mirrored z of view 1")
    z0 = planes[0][2]
    xt = np.clip(np.rint(np.arange(cols, dtype=np.float32)[None, :] - z0).astype(np.int64), 0, cols - 1)
    drs = np.take_along_axis(dr, xt, axis=1)
    keep = np.abs(z0 - drs) <= 1.0
    assert_same(dl, np.where(keep, z0, 0).astype(np.float32), "consistency mask")


@gpu
@pytest.mark.parametrize("f16", [0, 1])
def test_full_size_whole_frame_equals_the_definition(pm, oracle, synth, f16):
    """1280x720, 8 iterations, 11x11 (the shape of bench.py's plane legs), WHOLE frame, both maps, tolerance 0 against
    oracle/pm_planes_oracle.c on all host cores (the definition is per-pixel parallel by construction: its result does
    not depend on the thread count, tested above)."""
    import os
    import torch
    rows, cols = 720, 1280
    p = synth.make_pair(0, rows, cols)
    prm = pparams(pm, iters=8, f16=f16)
    L, R, _, _ = dev_pair([p])
    DL = torch.empty((1, rows, cols), dtype=torch.float32, device=L.device)
    DR = torch.empty_like(DL)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        e.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
        e.synchronize()
    kw = okw(prm)
    kw["nthreads"] = min(os.cpu_count() or 1, 16)
    el, er = oracle.planes_match(oracle.planes_params(**kw), p["left"], p["right"])
    assert_same(DL[0].cpu().numpy(), el, "whole frame, left map")
    assert_same(DR[0].cpu().numpy(), er, "whole frame, right map")


@gpu
def test_differential_fuzz_against_the_definition():
    """tools/fuzz_planes.py: random sizes / windows 3..15 / iteration counts / disparity ranges / slope and schedule
    constants / seeds / f32 and f16 state, whole Match() == oracle/pm_planes_oracle.c bit for bit (4000 cases were run
    after the last kernel change; 40 here)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_planes.py"), "--cases", "40", "--seed", "9"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@gpu
def test_configs4_workload_bgr_enhanced_f16_planes_full_size(pm, oracle, synth):
    """BASELINE configs[4] as ONE workload at its full shape: a 1280x720 BGR pair -> pm_stereo_ready of both images on
    the device (imaging::Normalize(NormalizeColorIlluminant(.)) -> gray, normalization.cpp:43-69,178-185) ->
    PM_MODE_PLANES with fp16 plane / cost state, 8 iterations, 11x11, through pm_match_device on the same stream, no
    host round trip.  Checks: run-to-run determinism; the enhanced gray images equal pm_enhance_oracle bit for bit at
    full size; the quality bar against the synthetic truth; and a 64-row full-width band of the enhanced pair, matched
    as its own problem by the engine, against pm_enhance_oracle o pm_planes_oracle bit for bit."""
    import torch
    import oracle_lib as O
    rows, cols = 720, 1280
    p = synth.make_pair(0, rows, cols)
    bl, br = synth.to_bgr(p["left"], 1), synth.to_bgr(p["right"], 2)
    prm = pparams(pm, iters=8, f16=1)
    dev = torch.device("cuda")
    BL, BR = torch.from_numpy(bl).to(dev).contiguous(), torch.from_numpy(br).to(dev).contiguous()
    GL = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    GR = torch.empty_like(GL)
    DL = torch.empty((1, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    with pm.Engine(prm, max_rows=rows, max_cols=cols) as e:
        def frame():
            e.stereo_ready(BL.data_ptr(), rows, cols, None, GL.data_ptr())
            e.stereo_ready(BR.data_ptr(), rows, cols, None, GR.data_ptr())
            e.match_device(1, GL.data_ptr(), GR.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
            e.synchronize()
        frame()
        first_l, first_r = DL.clone(), DR.clone()
        frame()
        assert torch.equal(first_l, DL) and torch.equal(first_r, DR), "configs[4] frame is not deterministic"
        # the same frame with the enhancement's per-pixel tail folded into the prep kernel (what bench.py times)
        FL, FR = torch.empty_like(DL), torch.empty_like(DR)
        e.match_bgr_device(1, BL.data_ptr(), BR.data_ptr(), rows, cols, None, None, FL.data_ptr(), FR.data_ptr())
        e.synchronize()
        assert torch.equal(FL, DL) and torch.equal(FR, DR), "pm_match_bgr_device differs from pm_stereo_ready + Match"
    gl, gr = GL.cpu().numpy(), GR.cpu().numpy()
    _, ol = O.stereo_ready(bl)
    _, orr = O.stereo_ready(br)
    assert np.array_equal(gl, ol) and np.array_equal(gr, orr), "stereo-ready gray images differ from pm_enhance_oracle"
    dl = DL[0].cpu().numpy()
    ok = dl > 0
    assert ok.mean() > 0.8
    assert (np.abs(dl - p["gt"])[ok] < 1).mean() >= 0.97
    # a 64-row full-width band of the ENHANCED pair as its own problem: engine (f16 state) == the mode's definition
    band = np.s_[328:392, :]
    with pm.Engine(prm, max_rows=64, max_cols=cols) as e:
        b_l, b_r = e.match(np.ascontiguousarray(gl[band]), np.ascontiguousarray(gr[band]))
    e_l, e_r = oracle.planes_match(oracle.planes_params(**okw(prm)), ol[band], orr[band])
    assert_same(b_l, e_l, "configs[4] band, left")
    assert_same(b_r, e_r, "configs[4] band, right")


@gpu
def test_batches_run_as_two_lanes_and_equal_their_singles(pm, synth):
    """A plane-mode batch advances its first half on the handle's stream and its second half on a side stream
    (pm_planes_host.hip::planes_match): five different pairs (3 + 2), with and without the device seeder, f32 and f16
    state -- every slot must equal that pair matched alone; and pm_match_bgr_device on a batch of three."""
    import torch
    rows, cols = 96, 256
    pairs = [synth.make_pair(60 + i, rows, cols, n_points=30, dilate_factor=2) for i in range(5)]
    for kw in (dict(iters=3, max_disp=48), dict(iters=2, max_disp=64, sparse_init=1, init_dilate_factor=2), dict(iters=2, max_disp=48, f16=1)):
        prm = pparams(pm, **kw)
        seeded = not kw.get("sparse_init")
        with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=5) as e:
            sl = [q["seed_l"] for q in pairs] if seeded else None
            sr = [q["seed_r"] for q in pairs] if seeded else None
            dls, drs = e.match_batch([q["left"] for q in pairs], [q["right"] for q in pairs], sl, sr)
            for i, q in enumerate(pairs):
                one = e.match(q["left"], q["right"], q["seed_l"] if seeded else None, q["seed_r"] if seeded else None)
                assert_same(dls[i], one[0], f"{kw}: slot {i} left")
                assert_same(drs[i], one[1], f"{kw}: slot {i} right")
    # BGR inputs, enhancement in the load path
    dev = torch.device("cuda")
    prm = pparams(pm, iters=2, f16=1)
    BL = torch.from_numpy(np.stack([synth.to_bgr(q["left"], 1) for q in pairs[:3]])).to(dev).contiguous()
    BR = torch.from_numpy(np.stack([synth.to_bgr(q["right"], 2) for q in pairs[:3]])).to(dev).contiguous()
    DL = torch.empty((3, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    OL = torch.empty((1, rows, cols), dtype=torch.float32, device=dev)
    OR = torch.empty_like(OL)
    with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=3) as e:
        e.match_bgr_device(3, BL.data_ptr(), BR.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
        e.synchronize()
        for i in range(3):
            e.match_bgr_device(1, BL[i].data_ptr(), BR[i].data_ptr(), rows, cols, None, None, OL.data_ptr(), OR.data_ptr())
            e.synchronize()
            assert torch.equal(DL[i], OL[0]) and torch.equal(DR[i], OR[0]), f"BGR batch slot {i}"


@gpu
def test_plane_mode_frames_of_a_sequence_run_in_pairs_and_equal_their_singles(pm, synth):
    """pm_submit_device in plane mode: while the device is busy a frame waits for its successor and the two run as one
    batch of two lanes (pm_hostpath.hip::enqueue_frames); whatever the grouping, every frame's maps are those of the
    pair matched alone.  Device-resident frames in neighbouring buffers (they can gang) and in scattered ones (they cannot)."""
    import torch
    rows, cols, depth = 96, 256, 4
    dev = torch.device("cuda")
    pairs = [synth.make_pair(70 + i, rows, cols, n_points=30, dilate_factor=2) for i in range(4)]
    prm = pparams(pm, iters=3, max_disp=48)
    L = torch.from_numpy(np.stack([q["left"] for q in pairs])).to(dev).contiguous()
    R = torch.from_numpy(np.stack([q["right"] for q in pairs])).to(dev).contiguous()
    DL = torch.empty((depth, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=depth) as e:
        singles = [e.match(q["left"], q["right"]) for q in pairs]
        def check(tag, pair_of, what):
            q = pair_of[tag]
            assert_same(DL[tag % depth].cpu().numpy(), singles[q][0], f"{what}: frame {tag} (pair {q}) left")
            assert_same(DR[tag % depth].cpu().numpy(), singles[q][1], f"{what}: frame {tag} (pair {q}) right")

        for order in ([0, 1, 2, 3], [2, 0, 3, 1]):  # inputs that are neighbours in memory / that are not
            pair_of = {}
            for i in range(12):
                if e.in_flight() == depth:  # the oldest frame's output slot is the one frame i writes
                    check(e.collect_device(), pair_of, f"order {order}")
                q = order[i % 4]
                pair_of[i] = q
                e.submit_device(L[q].data_ptr(), R[q].data_ptr(), rows, cols, None, None, DL[i % depth].data_ptr(),
                                DR[i % depth].data_ptr(), tag=i)
            while e.in_flight():
                check(e.collect_device(), pair_of, f"order {order}")

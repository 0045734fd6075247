"""Parity of the HIP engine (through the C ABI, via ctypes) against the CPU oracle.

Bar: bit-exact float32 disparity maps (tolerance 0) -- every arithmetic step of the engine is a
single IEEE rounding in the oracle's order, and the cost functor's sums are integers, so there is
no tolerance to state.  Cases follow the reference's own call patterns
(test/stereo_matching/patchmatch_test.cpp:149-183, patchmatch_gpu_test.cpp:68-88) on seeded
synthetic pairs at sizes the oracle finishes in seconds; full-size (1280x720, 8 iterations, 11x11)
coverage goes through size-independent properties and engine-vs-engine equality.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_same, small_pair

pytestmark = pytest.mark.gpu

SEMS = [0, 1]
ENGINES = [1, 2, 5]  # PM_ENGINE_SERIAL, _WAVE (anchors), _RUNBLK2 (product)


def mk(pm, sem, engine=0, patch=3, iters=3, lr=1, rows=64, cols=96, batch=1, **kw):
    p = pm.default_params(sem, patch=patch, patchmatch_iters=iters, engine=engine, left_right_check=lr, **kw)
    return pm.Engine(p, max_rows=rows, max_cols=cols, max_batch=batch)


def oparams(oracle, sem, patch=3, iters=3, lr=1, **kw):
    return oracle.default_params(sem, patch=patch, n_iters=iters, left_right_check=lr, nthreads=8, **kw)


# ---- single stages: one per reference function ------------------------------------------------------
@pytest.mark.parametrize("shape", [(8, 8), (37, 53), (64, 64), (48, 129)])
def test_gradient_magnitude(pm, oracle, shape):
    rng = np.random.default_rng(shape[0])
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    with mk(pm, 0, rows=shape[0], cols=shape[1]) as e:
        assert_same(e.gradient_magnitude(im), oracle.gradient_magnitude(im), "gradient")


def test_unit_noise_and_add_noise(pm, oracle):
    rows, cols = 45, 70
    with mk(pm, 1, rows=rows, cols=cols) as e:
        unit = e.unit_noise(rows, cols)
        assert_same(unit, oracle.rng_fill_uniform(rows * cols, -1.0, 1.0, 123).reshape(rows, cols), "unit noise")
        d = np.zeros((rows, cols), np.float32)
        d[10:30, 20:50] = 12.5
        d[0, 0] = 0.001
        for amp in (32.0, 8.0, 0.5, 0.0):
            got = e.add_noise(d, amp)
            assert_same(got, oracle.gpu_add_foreground_noise(d, unit, amp), f"AddForegroundNoise {amp}")
            assert_same(got, oracle.cpu_add_noise(d, amp, (d > 0).astype(np.uint8)), f"AddNoise {amp}")
        # the noise table follows the image size (the reference never resizes it)
        assert_same(e.unit_noise(20, 33), oracle.rng_fill_uniform(20 * 33, -1.0, 1.0, 123).reshape(20, 33), "resized")


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("pw,ph", [(3, 3), (5, 5), (7, 3), (3, 9), (11, 11), (15, 15)])
def test_cpu_propagate_each_pass(pm, oracle, synth, engine, pw, ph):
    rows, cols = 40, 72
    l, r, sl, _, _ = small_pair(synth, 11, rows, cols, n_points=20, dilate_factor=2)
    ims = oracle.ImageSet(l, r)
    d = oracle.cpu_add_noise(sl, 8.0, (sl > 0).astype(np.uint8))
    with mk(pm, 0, engine, rows=rows, cols=cols) as e:
        for mask in (1, 2, 4, 8, 15):
            got = e.propagate(l, r, d, ph, pw, mask)
            assert_same(got, oracle.cpu_propagate(ims, d, ph, pw, pass_mask=mask, nthreads=8), f"pass mask {mask}")


@pytest.mark.parametrize("engine", ENGINES)
def test_gpu_propagate_each_sweep(pm, oracle, synth, engine):
    rows, cols = 40, 72
    l, r, sl, _, _ = small_pair(synth, 12, rows, cols, n_points=20, dilate_factor=2)
    ims = oracle.ImageSet(l, r)
    unit = oracle.rng_fill_uniform(rows * cols, -1.0, 1.0).reshape(rows, cols)
    d = oracle.gpu_add_foreground_noise(sl, unit, 16.0)
    with mk(pm, 1, engine, rows=rows, cols=cols) as e:
        for mask in (1, 2, 4, 8, 15):
            assert_same(e.propagate(l, r, d, 3, 3, mask), oracle.gpu_propagate(ims, d, pass_mask=mask, nthreads=8),
                        f"sweep mask {mask}")


@pytest.mark.parametrize("engine", [2, 5])
@pytest.mark.parametrize("kind", ["random", "plateaus", "huge", "tiny"])
def test_gpu_propagate_adversarial_fields(pm, oracle, synth, engine, kind):
    """PM_SEM_GPU sweeps on disparity fields that exercise the clamp (x - d < 1), the single-position slow
    path, long runs of one value and binade crossings of the sample positions."""
    rows, cols = 45, 333
    l, r, _, _, _ = small_pair(synth, 21, rows, cols, n_points=20, dilate_factor=2)
    ims = oracle.ImageSet(l, r)
    rng = np.random.default_rng(5)
    if kind == "random":
        d = rng.uniform(0.0, 90.0, (rows, cols)).astype(np.float32)
    elif kind == "plateaus":
        d = np.repeat(np.repeat(rng.uniform(0.0, 40.0, (rows // 5 + 1, cols // 9 + 1)), 5, 0), 9, 1)
        d = d[:rows, :cols].astype(np.float32)
        d[rng.random((rows, cols)) < 0.05] = 0.0
    elif kind == "huge":  # mostly clamped candidates
        d = rng.uniform(100.0, 600.0, (rows, cols)).astype(np.float32)
        d[:, ::7] = 3.25
    else:  # values around powers of two of x - d, and denormal-small disparities
        xs = np.arange(cols, dtype=np.float32)[None, :].repeat(rows, 0)
        pick = rng.choice(np.array([1, 2, 4, 8, 16, 32, 64, 128], np.float32), (rows, cols))
        d = np.maximum(xs - pick + rng.choice(np.array([-1e-3, 0, 1e-3, 0.5], np.float32), (rows, cols)), 0)
        d[rng.random((rows, cols)) < 0.1] = 1e-30
        d = d.astype(np.float32)
    with mk(pm, 1, engine, rows=rows, cols=cols) as e:
        for mask in (1, 2, 4, 8, 15):
            assert_same(e.propagate(l, r, d, 3, 3, mask), oracle.gpu_propagate(ims, d, pass_mask=mask, nthreads=8),
                        f"{kind} sweep mask {mask}")


@pytest.mark.parametrize("engine", [1, 5])
@pytest.mark.parametrize("pw", [5, 11])
def test_cpu_propagate_lerp_weight_extremes(pm, oracle, synth, engine, pw):
    """PM_SEM_CPU sweeps with sample positions within a few ulps of an integer column: the fixed-point lerp weights
    reach 0 and 65536 (the run engine packs them into 16 bits for v_dot2_u32_u16 and relies on the other weight
    being 0 then), on both sides of the integer, next to ordinary fractions and long runs."""
    rows, cols = 40, 200
    l, r, _, _, _ = small_pair(synth, 23, rows, cols, n_points=20, dilate_factor=2)
    ims = oracle.ImageSet(l, r)
    rng = np.random.default_rng(17)
    xs = np.arange(cols, dtype=np.float32)[None, :].repeat(rows, 0)
    shift = np.float32((pw - 1) * 0.5)
    # x - d - shift = k + eps  with  eps in {0, +-1 ulp ... +-2^-16, +-2^-17, +-2^-18}
    k = rng.integers(0, 60, (rows, cols)).astype(np.float32)
    eps = rng.choice(np.array([0.0, 2.0 ** -16, -2.0 ** -16, 2.0 ** -17, -2.0 ** -17, 2.0 ** -18, -2.0 ** -18, 7.6e-6,
                               -7.6e-6, 7.7e-6, -7.7e-6, 0.25, 0.5], np.float32), (rows, cols))
    d = np.maximum(xs - shift - k - eps, 0).astype(np.float32)
    d = np.repeat(d[:, ::3], 3, axis=1)[:, :cols]        # runs of three equal values
    d[rng.random((rows, cols)) < 0.3] = np.float32(13.99999)
    with mk(pm, 0, engine, rows=rows, cols=cols) as e:
        for mask in (1, 2, 4, 8, 15):
            assert_same(e.propagate(l, r, d, pw, pw, mask), oracle.cpu_propagate(ims, d, pw, pw, pass_mask=mask, nthreads=8),
                        f"weight extremes, pass mask {mask}")


@pytest.mark.parametrize("engine", [1, 5])
@pytest.mark.parametrize("pw", [5, 11])
@pytest.mark.parametrize("layout", ["rows", "cols"])
def test_cpu_propagate_runs_through_every_segment(pm, oracle, engine, pw, layout):
    """PM_SEM_CPU sweeps in which ONE value runs along whole chains: the right image is the left one shifted by 6 pixels, the
    map is wrong everywhere except for one pixel per chain that holds 6.0 -- at the chain's first position, somewhere in
    the middle, or not at all -- and stretches without texture stop the run (every candidate costs the same there, and
    the rule is a strict <).  The run engine hands such a run from segment to segment in its fix-up rounds, one round per
    segment, and since round 6 a re-run that is alone in its wavefront goes wide (run3_step<WIDE>: two or four groups on one
    run; windows of 5 take 16-lane groups, 11 takes 32 here); chains of 890 positions are 8 / 16 segments."""
    rng = np.random.default_rng(23)
    long_, short = 900, 22
    rows, cols = (short, long_) if layout == "rows" else (long_, short + 30)
    l = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
    if layout == "rows":
        l[14:, 300:330] = 77          # no texture: the run stops here in the lower rows
    else:
        l[400:430, 30:] = 77
    r = np.roll(l, -6, axis=1)        # left (x) = right (x - 6)
    d = rng.uniform(8.0, 40.0, (rows, cols)).astype(np.float32)
    h = pw // 2
    if layout == "rows":
        d[:8, h] = 6.0                # from the first position of the forward sweep
        d[4:12, cols - 1 - h] = 6.0   # ... and of the backward sweep
        for y in range(12, rows):
            d[y, rng.integers(h, cols - h)] = 6.0
    else:
        d[h, 7:25] = 6.0
        d[rows - 1 - h, 20:40] = 6.0
        for x in range(40, cols):
            d[rng.integers(h, rows - h), x] = 6.0
    ims = oracle.ImageSet(l, r)
    with mk(pm, 0, engine, rows=rows, cols=cols) as e:
        for mask in (1, 2, 4, 8, 15):
            want = oracle.cpu_propagate(ims, d, pw, pw, pass_mask=mask, nthreads=8)
            assert_same(e.propagate(l, r, d, pw, pw, mask), want, f"{layout} pass mask {mask}")
            if mask in (1, 2) and layout == ("rows" if mask == 1 else "cols"):
                # the test is what it claims to be: the value did run far
                assert (want == 6.0).mean() > 0.2


def test_remove_background_and_mask_occlusions(pm, oracle, synth):
    rows, cols = 50, 90
    l, r, sl, sr, _ = small_pair(synth, 13, rows, cols, n_points=30, dilate_factor=2)
    ims = oracle.ImageSet(l, r)
    d = oracle.cpu_add_noise(sl, 2.0, (sl > 0).astype(np.uint8))
    with mk(pm, 0, rows=rows, cols=cols) as e:
        for (pw, ph, f) in ((3, 3, 1.5), (5, 5, 2.0), (11, 7, 1.5)):
            assert_same(e.remove_background(l, r, d, ph, pw, f), oracle.cpu_remove_background(ims, d, ph, pw, f),
                        f"RemoveBackground {pw}x{ph}")
        assert_same(e.mask_occlusions(sl, sr), oracle.gpu_mask_occlusions(sl, sr), "MaskOcclusions")
    with mk(pm, 1, rows=rows, cols=cols) as e:
        for f in (0.8, 0.5):
            assert_same(e.remove_background(l, r, d, 3, 3, f), oracle.gpu_mask_background(ims, d, 0.9, f),
                        f"MaskBackground {f}")


# ---- the whole path ------------------------------------------------------------------------------------
@pytest.mark.parametrize("sem", SEMS)
@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("rows,cols", [(48, 80), (67, 131), (120, 188), (61, 99), (62, 100)])  # rows: every remainder mod 4
def test_match_both_views(pm, oracle, synth, sem, engine, rows, cols):
    l, r, sl, sr, _ = small_pair(synth, rows, rows, cols, n_points=40, dilate_factor=2)
    with mk(pm, sem, engine, patch=5, iters=3, rows=rows, cols=cols) as e:
        dl, dr = e.match(l, r, sl, sr)
    el, er = oracle.match(oparams(oracle, sem, 5, 3), l, r, sl, sr)
    assert_same(dl, el, "left disparity")
    assert_same(dr, er, "right disparity")
    assert (dl > 0).mean() > 0.2  # the case is not degenerate


@pytest.mark.parametrize("patch", [3, 7, 11])
def test_match_cpu_semantics_window_sizes(pm, oracle, synth, patch):
    rows, cols = 60, 100
    l, r, sl, sr, _ = small_pair(synth, 20 + patch, rows, cols, n_points=40, dilate_factor=2)
    with mk(pm, 0, 0, patch=patch, iters=4, rows=rows, cols=cols) as e:
        dl, dr = e.match(l, r, sl, sr)
    el, er = oracle.match(oparams(oracle, 0, patch, 4), l, r, sl, sr)
    assert_same(dl, el, "left")
    assert_same(dr, er, "right")


def test_match_reference_cpu_test_recipe(pm, oracle, synth):
    # patchmatch_test.cpp:173-183: noise 32, 8, 2, 0.5; windows 5x5, 5x5, 3x3, 3x3; background 3x3 / 1.5
    rows, cols = 120, 188  # 376x240 halved, the reference test size / 2
    l, r, sl, sr, _ = small_pair(synth, 31, rows, cols, n_points=60, dilate_factor=3)
    sched = dict(noise_amp=[32.0, 8.0, 2.0, 0.5], patch_w=[5, 5, 3, 3], patch_h=[5, 5, 3, 3])
    p = pm.default_params(0, patchmatch_iters=4, bg_patch_w=3, bg_patch_h=3, win_by_factor=1.5, left_right_check=0,
                          **sched)
    with pm.Engine(p, max_rows=rows, max_cols=cols) as e:
        dl, _ = e.match(l, r, sl, None)
    op = oracle.default_params(0, n_iters=4, bg_patch_w=3, bg_patch_h=3, bg_factor=1.5, left_right_check=0,
                               nthreads=8, literal=1, **sched)
    el, _ = oracle.match(op, l, r, sl, None)
    assert_same(dl, el, "patchmatch_test.cpp recipe (literal oracle)")


def test_match_reference_gpu_test_settings(pm, oracle, synth):
    # patchmatch_gpu_test.cpp:68-88: 376x240, alpha 0.9, 3 iterations, Match called 5 times in a row
    rows, cols = 240, 376
    l, r, sl, sr, _ = small_pair(synth, 32, rows, cols)
    el, er = oracle.match(oparams(oracle, 1, 3, 3), l, r, sl, sr)
    with mk(pm, 1, rows=rows, cols=cols) as e:
        for _ in range(5):
            dl, dr = e.match(l, r, sl, sr)
            assert_same(dl, el, "left")
            assert_same(dr, er, "right")


@pytest.mark.parametrize("sem", SEMS)
def test_batch_equals_singles(pm, oracle, synth, sem):
    rows, cols = 40, 70
    pairs = [small_pair(synth, 40 + i, rows, cols, n_points=20, dilate_factor=2) for i in range(3)]
    with mk(pm, sem, patch=5, iters=2, rows=rows, cols=cols, batch=3) as e:
        dls, drs = e.match_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs],
                                 [p[3] for p in pairs])
    for i, p in enumerate(pairs):
        el, er = oracle.match(oparams(oracle, sem, 5, 2), p[0], p[1], p[2], p[3])
        assert_same(dls[i], el, f"pair {i} left")
        assert_same(drs[i], er, f"pair {i} right")


@pytest.mark.parametrize("sem,patch", [(0, 11), (0, 3), (1, 3)])
def test_batch_pipelines_on_lanes_equal_singles(pm, oracle, synth, sem, patch):
    """A batch of more than two pairs runs as pipelines of two pairs that take two lanes of view streams in rotation
    (pm_engine.hip::run_pairs_on_lanes): 5 pairs = pipelines {0,1} {2,3} {4}.  Every slot must equal the Match of its
    pair alone, twice in a row (the lanes are re-used), and -- recorded as ONE HIP graph -- on replay with new inputs."""
    torch = pytest.importorskip("torch")
    rows, cols, nb = 72, 100, 5
    dev = torch.device("cuda:0")
    sets = [[small_pair(synth, 300 + 10 * k + i, rows, cols, n_points=25, dilate_factor=2) for i in range(nb)]
            for k in range(2)]
    st = lambda prs, j, dt: torch.from_numpy(np.stack([p[j] for p in prs])).to(dev, dt).contiguous()
    bufs = [st(sets[0], 0, torch.uint8), st(sets[0], 1, torch.uint8), st(sets[0], 2, torch.float32),
            st(sets[0], 3, torch.float32)]
    DL = torch.empty((nb, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    singles = []
    with mk(pm, sem, patch=patch, iters=3, rows=rows, cols=cols) as e:
        for prs in sets:
            singles.append([e.match(p[0], p[1], p[2], p[3]) for p in prs])
    with mk(pm, sem, patch=patch, iters=3, rows=rows, cols=cols, batch=nb) as e:
        run = lambda: e.match_device(nb, bufs[0].data_ptr(), bufs[1].data_ptr(), rows, cols, bufs[2].data_ptr(),
                                     bufs[3].data_ptr(), DL.data_ptr(), DR.data_ptr())
        for rep in range(2):
            DL.zero_(); DR.zero_()
            torch.cuda.synchronize()
            run()
            e.synchronize()
            for i in range(nb):
                assert_same(DL[i].cpu().numpy(), singles[0][i][0], f"call {rep}, slot {i}, left")
                assert_same(DR[i].cpu().numpy(), singles[0][i][1], f"call {rep}, slot {i}, right")
        e.capture_begin()
        run()
        e.capture_end()
        for j, dt in ((0, torch.uint8), (1, torch.uint8), (2, torch.float32), (3, torch.float32)):
            bufs[j].copy_(st(sets[1], j, dt))
        torch.cuda.synchronize()
        e.replay()
        e.synchronize()
        for i in range(nb):
            assert_same(DL[i].cpu().numpy(), singles[1][i][0], f"replay, slot {i}, left")
            assert_same(DR[i].cpu().numpy(), singles[1][i][1], f"replay, slot {i}, right")
    el, er = oracle.match(oparams(oracle, sem, patch, 3), *sets[1][4][:4])
    assert_same(singles[1][4][0], el, "the single-pair anchor against the oracle, left")
    assert_same(singles[1][4][1], er, "right")


@pytest.mark.parametrize("sem", SEMS)
def test_self_seeded_batch_on_lanes_equals_singles(pm, oracle, synth, sem):
    """sparse_init with no seed maps: every pipeline of a batch seeds its views on the device, with the seeder scratch of
    its own lane and view (pm_handle::seeds) -- lanes run side by side."""
    torch = pytest.importorskip("torch")
    rows, cols, nb = 96, 160, 5
    dev = torch.device("cuda:0")
    prs = [small_pair(synth, 500 + i, rows, cols, n_points=25, dilate_factor=2) for i in range(nb)]
    L = torch.from_numpy(np.stack([p[0] for p in prs])).to(dev).contiguous()
    R = torch.from_numpy(np.stack([p[1] for p in prs])).to(dev).contiguous()
    DL = torch.empty((nb, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    kw = dict(sparse_init=1, max_features_per_frame=60, min_distance_btw_features=8, max_disp=48, templ_cols=15,
              templ_rows=7)
    with mk(pm, sem, patch=3, iters=2, rows=rows, cols=cols, **kw) as e:
        singles = [e.match(p[0], p[1], None, None) for p in prs]
    assert any((s[0] > 0).any() for s in singles), "the seeder found nothing: the test would compare zeros"
    with mk(pm, sem, patch=3, iters=2, rows=rows, cols=cols, batch=nb, **kw) as e:
        for rep in range(2):
            e.match_device(nb, L.data_ptr(), R.data_ptr(), rows, cols, None, None, DL.data_ptr(), DR.data_ptr())
            e.synchronize()
            for i in range(nb):
                assert_same(DL[i].cpu().numpy(), singles[i][0], f"call {rep}, slot {i}, left")
                assert_same(DR[i].cpu().numpy(), singles[i][1], f"call {rep}, slot {i}, right")


# ---- edge cases ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("depth", [1, 3])
def test_pipelined_sequence_equals_single_matches(pm, oracle, synth, depth):
    """pm_submit_u8 / pm_collect: results of pm_match_u8, in submission order, with <= max_batch in flight."""
    rows, cols = 60, 100
    pairs = [small_pair(synth, 40 + i, rows, cols, n_points=25, dilate_factor=2) for i in range(5)]
    params = pm.default_params(0, patch=5, patchmatch_iters=2)
    with pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=1) as e:
        want = [e.match(l, r, sl, sr) for (l, r, sl, sr, _) in pairs]
    got = []
    with pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=depth) as e:
        with pytest.raises(pm.PmError) as err:
            e.collect()
        assert err.value.status == pm.PM_ERR_BUSY
        for i, (l, r, sl, sr, _) in enumerate(pairs):
            if e.in_flight() == depth:
                with pytest.raises(pm.PmError) as err:
                    e.submit(l, r, sl, sr, tag=99)
                assert err.value.status == pm.PM_ERR_BUSY
                got.append(e.collect())
            e.submit(l, r, sl, sr, tag=100 + i)
        while e.in_flight():
            got.append(e.collect())
    assert [t for (_, _, t) in got] == [100 + i for i in range(5)]
    for i, ((dl, dr, _), (wl, wr)) in enumerate(zip(got, want)):
        assert_same(dl, wl, f"pair {i} left")
        assert_same(dr, wr, f"pair {i} right")


@pytest.mark.parametrize("sem,patch,mode", [(0, 5, 0), (1, 3, 0), (0, 5, 1)])
def test_sequence_on_page_locked_memory_with_bound_maps(pm, oracle, synth, sem, patch, mode):
    """pm_host_alloc + pm_submit_bound_u8: images read and maps written in place by DMA, frames overlapping on the
    device as chunks of one or two (held frames, pm_flush), results those of pm_match_u8 in submission order; a mix of
    page-locked and pageable buffers; the plane mode takes the one-frame-after-the-other route."""
    rows, cols = 60, 100
    pairs = [small_pair(synth, 140 + i, rows, cols, n_points=25, dilate_factor=2) for i in range(9)]
    kw = dict(mode=pm.PM_MODE_PLANES, max_disp=48) if mode else {}
    params = pm.default_params(sem, patch=patch, patchmatch_iters=2, **kw)
    with pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=1) as e:
        want = [e.match(l, r, sl, sr) for (l, r, sl, sr, _) in pairs]
    with pm.Engine(params, max_rows=rows + 4, max_cols=cols + 8, max_batch=4) as e:
        ins = []
        for i, (l, r, sl, sr, _) in enumerate(pairs):
            row = []
            for a in (l, r, sl, sr):
                if i % 4 == 3:          # every fourth frame comes from pageable memory
                    row.append(a)
                else:
                    b = e.host_alloc(a.shape, a.dtype)
                    np.copyto(b, a)
                    row.append(b)
            ins.append(row)
        outs = [(e.host_alloc((rows, cols), np.float32), e.host_alloc((rows, cols), np.float32)) for _ in range(3)]
        outs.append((np.empty((rows, cols), np.float32), np.empty((rows, cols), np.float32)))  # pageable maps
        got, tags, done = [], [], 0
        for i in range(len(pairs)):
            if e.in_flight() == 4:
                dl, dr, tag = e.collect()
                assert dl is outs[done % 4][0]
                got.append((dl.copy(), dr.copy())); tags.append(tag); done += 1
            e.submit(*ins[i], tag=500 + i, out=outs[i % 4])
            if i == 5:
                e.flush()
        with pytest.raises(pm.PmError) as err:   # other maps than the bound ones
            e.collect(out=(np.empty((rows, cols), np.float32), np.empty((rows, cols), np.float32)))
        assert err.value.status == pm.PM_ERR_INVALID_ARG
        with pytest.raises(pm.PmError) as err:   # the memory may be the target of a DMA
            e.host_free(outs[0][0])
        assert err.value.status == pm.PM_ERR_BUSY
        with pytest.raises(pm.PmError) as err:   # frames in flight live in the plane slots a device match would use
            e.match(pairs[0][0], pairs[0][1])
        assert err.value.status == pm.PM_ERR_BUSY
        while e.in_flight():
            dl, dr, tag = e.collect()
            got.append((dl.copy(), dr.copy())); tags.append(tag); done += 1
        e.host_free(outs[0][0])
        # a registered range: plain numpy memory, page-locked in place
        reg = np.empty((rows, cols), np.float32)
        regr = np.empty((rows, cols), np.float32)
        e.host_register(reg); e.host_register(regr)
        l, r, sl, sr, _ = pairs[2]
        e.submit(l, r, sl, sr, tag=7, out=(reg, regr))
        e.collect()
        assert_same(reg, want[2][0], "registered map, left")
        assert_same(regr, want[2][1], "registered map, right")
        e.host_unregister(reg); e.host_unregister(regr)
        with pytest.raises(pm.PmError):
            e.host_unregister(reg)
    assert tags == [500 + i for i in range(len(pairs))]
    for i, ((dl, dr), (wl, wr)) in enumerate(zip(got, want)):
        assert_same(dl, wl, f"pair {i} left")
        assert_same(dr, wr, f"pair {i} right")


def test_device_resident_sequence_and_self_seeded_frames(pm, oracle, synth):
    """pm_submit_device: nothing is copied, frames that lie next to each other in device memory run as one chunk;
    with sparse_init every frame seeds itself on the view streams' seeder scratch."""
    torch = pytest.importorskip("torch")
    rows, cols = 96, 160
    dev = torch.device("cuda:0")
    n = 6
    pairs = [small_pair(synth, 160 + i, rows, cols, n_points=25, dilate_factor=2) for i in range(n)]
    params = pm.default_params(0, patch=5, patchmatch_iters=2, sparse_init=1)
    with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
        want = [e.match(p[0], p[1]) for p in pairs]
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    DL = torch.full((n, rows, cols), -1.0, dtype=torch.float32, device=dev)
    DR = torch.full_like(DL, -1.0)
    torch.cuda.synchronize()
    with pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=4) as e:
        tags = []
        for i in range(n):
            if e.in_flight() == 4:
                tags.append(e.collect_device())
            e.submit_device(L[i].data_ptr(), R[i].data_ptr(), rows, cols, None, None, DL[i].data_ptr(), DR[i].data_ptr(),
                            tag=i)
        while e.in_flight():
            tags.append(e.collect_device())
    assert tags == list(range(n))
    for i in range(n):
        assert_same(DL[i].cpu().numpy(), want[i][0], f"frame {i} left")
        assert_same(DR[i].cpu().numpy(), want[i][1], f"frame {i} right")


@pytest.mark.parametrize("self_seed", [0, 1])
def test_device_sequence_waits_for_the_producers_event(pm, synth, self_seed):
    """pm_submit_device_after: inputs still being written on the device -- here by torch copies on a side stream, behind a
    long chain of filler kernels, with NO host synchronisation -- are complete before either view (the second runs on an
    internal stream) or the self-seeding head reads them: the frame waits for the caller's event on the device."""
    torch = pytest.importorskip("torch")
    rows, cols = 96, 160
    dev = torch.device("cuda:0")
    n = 4
    pairs = [small_pair(synth, 170 + i, rows, cols, n_points=25, dilate_factor=2) for i in range(n)]
    params = pm.default_params(0, patch=5, patchmatch_iters=2, sparse_init=self_seed)
    with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
        want = [e.match(p[0], p[1], None if self_seed else p[2], None if self_seed else p[3]) for p in pairs]
    src = {k: torch.from_numpy(np.stack([p[j] for p in pairs])).to(dev) for j, k in enumerate(("l", "r", "sl", "sr"))}
    L, R = torch.zeros_like(src["l"]), torch.zeros_like(src["r"])          # garbage until the producer has run
    SL, SR = torch.zeros_like(src["sl"]), torch.zeros_like(src["sr"])
    DL = torch.full((n, rows, cols), -1.0, dtype=torch.float32, device=dev)
    DR = torch.full_like(DL, -1.0)
    filler = torch.zeros((4096, 4096), dtype=torch.float32, device=dev)
    producer = torch.cuda.Stream()
    events = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    with pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=4) as e:
        for i in range(n):
            with torch.cuda.stream(producer):
                for _ in range(20):          # ~ a millisecond of work in front of the inputs
                    filler.add_(1.0)
                L[i].copy_(src["l"][i]); R[i].copy_(src["r"][i])
                SL[i].copy_(src["sl"][i]); SR[i].copy_(src["sr"][i])
                events[i].record(producer)
            e.submit_device(L[i].data_ptr(), R[i].data_ptr(), rows, cols, None if self_seed else SL[i].data_ptr(),
                            None if self_seed else SR[i].data_ptr(), DL[i].data_ptr(), DR[i].data_ptr(), tag=i,
                            ready_event=events[i].cuda_event)
        while e.in_flight():
            e.collect_device()
    for i in range(n):
        assert_same(DL[i].cpu().numpy(), want[i][0], f"frame {i} left")
        assert_same(DR[i].cpu().numpy(), want[i][1], f"frame {i} right")


def test_collect_starts_a_held_frame(pm, synth):
    """The loop submit(k + 1); collect(k) at depth 2: frame k + 1 is held while the device is busy with frame k, and
    pm_collect(k) enqueues it once k is through -- in_flight frames never sit idle until the next call."""
    import time
    rows, cols = 720, 1280
    p = synth.make_pair(3, rows, cols)
    params = pm.default_params(0, patch=11, patchmatch_iters=8)
    with pm.Engine(params, max_rows=rows, max_cols=cols, max_batch=3) as e:
        a = e.match(p["left"], p["right"], p["seed_l"], p["seed_r"])
        # maps bound in page-locked memory: pm_collect is the event wait and nothing else
        outs = [(e.host_alloc((rows, cols), np.float32), e.host_alloc((rows, cols), np.float32)) for _ in range(2)]
        for o in outs:
            o[0].fill(-1.0)
            o[1].fill(-1.0)
        e.submit(p["left"], p["right"], p["seed_l"], p["seed_r"], tag=0, out=outs[0])
        e.submit(p["left"], p["right"], p["seed_l"], p["seed_r"], tag=1, out=outs[1])  # held: the device is busy with frame 0
        e.collect()                         # ... and started by this call, once frame 0 is through
        # Frame 1 (~2.5 ms of device work) finishes without any further call: its maps, bound in page-locked memory and
        # written in place by the device, fill up while the host only watches (no clock is asserted on: a held frame that
        # nobody started would leave the -1 fill for ever)
        deadline = time.perf_counter() + 5.0
        done = False
        while not done and time.perf_counter() < deadline:
            done = np.array_equal(outs[1][0], a[0]) and np.array_equal(outs[1][1], a[1])
            time.sleep(0.002)
        assert done, "the held frame was not started by the collect of the frame in front of it"
        e.collect()
        for o in outs:
            assert_same(o[0], a[0], "left")
            assert_same(o[1], a[1], "right")


def test_edges_no_seeds_one_view_strides_and_errors(pm, oracle, synth):
    rows, cols = 33, 47
    l, r, sl, sr, _ = small_pair(synth, 50, rows, cols, n_points=12, dilate_factor=2)
    with mk(pm, 0, patch=3, iters=2, rows=64, cols=64) as e:   # plan larger than the image
        dl, dr = e.match(l, r, None, None)
        assert not dl.any() and not dr.any()
        dl, dr = e.match(l, r, sl, None)
        el, er = oracle.match(oparams(oracle, 0, 3, 2), l, r, sl, None)
        assert_same(dl, el)
        assert_same(dr, er)
        # row strides: images embedded in wider buffers (cv::Mat::step)
        lw = np.zeros((rows, cols + 13), np.uint8); lw[:, :cols] = l
        rw = np.zeros((rows, cols + 13), np.uint8); rw[:, :cols] = r
        out_l = np.full((rows, cols + 5), -1, np.float32)
        out_r = np.full((rows, cols + 5), -1, np.float32)
        slc = np.ascontiguousarray(sl)
        rc = e.lib.pm_match_u8(e.h, lw.ctypes.data, rw.ctypes.data, rows, cols, cols + 13, slc.ctypes.data, None, 0,
                               out_l.ctypes.data, out_r.ctypes.data, 4 * (cols + 5))
        assert rc == 0
        assert_same(out_l[:, :cols], el, "strided left")
        assert np.all(out_l[:, cols:] == -1)
        # errors: too large for the plan, null pointers, tiny image
        big = np.zeros((65, 64), np.uint8)
        with pytest.raises(pm.PmError) as ex:
            e.match(big, big)
        assert ex.value.status == pm.PM_ERR_SIZE
        assert e.lib.pm_match_u8(e.h, None, rw.ctypes.data, rows, cols, 0, None, None, 0, out_l.ctypes.data,
                                 out_r.ctypes.data, 0) == pm.PM_ERR_INVALID_ARG
        with pytest.raises(pm.PmError):
            e.match(np.zeros((4, 4), np.uint8), np.zeros((4, 4), np.uint8))
        with pytest.raises(pm.PmError):
            e.propagate(l, r, sl, 4, 3)
        # smallest supported image, and the engine still works after the errors
        t = np.arange(64, dtype=np.uint8).reshape(8, 8) * 3
        s8 = np.full((8, 8), 2.0, np.float32)
        dl, dr = e.match(t, t, s8, s8)
        el, er = oracle.match(oparams(oracle, 0, 3, 2), t, t, s8, s8)
        assert_same(dl, el)
        assert_same(dr, er)


def test_left_right_check_off(pm, oracle, synth):
    rows, cols = 40, 64
    l, r, sl, sr, _ = small_pair(synth, 51, rows, cols, n_points=20, dilate_factor=2)
    for sem in SEMS:
        with mk(pm, sem, patch=5, iters=2, lr=0, rows=rows, cols=cols) as e:
            dl, dr = e.match(l, r, sl, sr)
        assert dr is None
        el, _ = oracle.match(oparams(oracle, sem, 5, 2, lr=0), l, r, sl, sr)
        assert_same(dl, el)


def test_device_entry_point_with_torch_buffers(pm, oracle, synth):
    torch = pytest.importorskip("torch")
    rows, cols = 48, 80
    pairs = [small_pair(synth, 60 + i, rows, cols, n_points=20, dilate_factor=2) for i in range(2)]
    dev = torch.device("cuda:0")
    t = lambda k, dt: torch.from_numpy(np.stack([p[k] for p in pairs])).to(dev, dt).contiguous()
    L, R, SL, SR = t(0, torch.uint8), t(1, torch.uint8), t(2, torch.float32), t(3, torch.float32)
    DL = torch.empty((2, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    torch.cuda.synchronize()
    with mk(pm, 0, patch=5, iters=2, rows=rows, cols=cols, batch=2) as e:
        e.match_device(2, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(),
                       DR.data_ptr())
        e.synchronize()
        prof_off = e.profile_read()
        assert sum(v[0] for v in prof_off.values()) == 0
        e.profile_enable(True)
        e.match_device(2, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(),
                       DR.data_ptr())
        prof = e.profile_read()
    # 2 iterations x 2 row sweeps (and column sweeps), x 2 when the views run on their own streams (default), x 2 when
    # the two pairs run as pipelines of their own (PM_PAIR_CHUNK=1)
    assert prof["sweep_row"][0] in (4, 8, 16) and prof["sweep_col"][0] == prof["sweep_row"][0]
    assert prof["noise_cost"][0] == prof["sweep_row"][0] // 2
    assert prof["sweep_row"][1] > 0
    for i, p in enumerate(pairs):
        el, er = oracle.match(oparams(oracle, 0, 5, 2), p[0], p[1], p[2], p[3])
        assert_same(DL[i].cpu().numpy(), el, f"pair {i} left")
        assert_same(DR[i].cpu().numpy(), er, f"pair {i} right")


@pytest.mark.parametrize("sem", SEMS)
@pytest.mark.parametrize("rows,cols", [(24, 6000), (6000, 24), (20, 11000)])
def test_very_long_chains(pm, oracle, synth, sem, rows, cols):
    """Chains longer than 64 KB of LDS state (raised per-kernel limit) and longer than the CU's LDS (serial
    fallback): still the oracle's maps.  Long axis 6000: 96 KB per chain; 11000: beyond 160 KB."""
    rng = np.random.default_rng(rows + cols)
    base = rng.integers(0, 256, (rows, cols + 40), dtype=np.uint8)
    left, right = np.ascontiguousarray(base[:, 20:20 + cols]), np.ascontiguousarray(base[:, 14:14 + cols])  # d = 6
    seed = np.zeros((rows, cols), np.float32)
    seed[rows // 2 - 2:rows // 2 + 3, ::97] = 6.0
    seed_r = seed.copy()
    with mk(pm, sem, patch=5, iters=2, rows=rows, cols=cols) as e:
        dl, dr = e.match(left, right, seed, seed_r)
    el, er = oracle.match(oparams(oracle, sem, 5, 2), left, right, seed, seed_r)
    assert_same(dl, el, "left")
    assert_same(dr, er, "right")


def test_graph_replay_equals_direct_calls(pm, oracle, synth):
    """pm_capture_begin / pm_capture_end / pm_replay: the recorded Match() (both view streams) as one HIP graph."""
    torch = pytest.importorskip("torch")
    rows, cols = 64, 96
    dev = torch.device("cuda:0")
    pa = small_pair(synth, 91, rows, cols, n_points=25, dilate_factor=2)
    pb = small_pair(synth, 92, rows, cols, n_points=25, dilate_factor=2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    L, R, SL, SR = t(pa[0]), t(pa[1]), t(pa[2]), t(pa[3])
    DL, DR = torch.empty_like(SL), torch.empty_like(SR)
    with mk(pm, 0, patch=5, iters=2, rows=rows, cols=cols) as e:
        run = lambda: e.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(),
                                     DL.data_ptr(), DR.data_ptr())
        with pytest.raises(pm.PmError):
            e.replay()  # nothing captured yet
        run()
        e.synchronize()
        e.capture_begin()
        run()
        e.capture_end()
        # new inputs in the same buffers: the replay works on what the buffers hold now
        L.copy_(t(pb[0])); R.copy_(t(pb[1])); SL.copy_(t(pb[2])); SR.copy_(t(pb[3]))
        torch.cuda.synchronize()
        e.replay()
        e.synchronize()
    el, er = oracle.match(oparams(oracle, 0, 5, 2), pb[0], pb[1], pb[2], pb[3])
    assert_same(DL.cpu().numpy(), el, "replayed left")
    assert_same(DR.cpu().numpy(), er, "replayed right")


def test_capture_error_paths_leave_the_handle_usable(pm, oracle, synth):
    """A call that fails while capturing ends the capture; calls that would synchronise or record events are
    refused with PM_ERR_BUSY between pm_capture_begin and pm_capture_end."""
    torch = pytest.importorskip("torch")
    rows, cols = 64, 96
    dev = torch.device("cuda:0")
    pa = small_pair(synth, 93, rows, cols, n_points=25, dilate_factor=2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    L, R, SL, SR = t(pa[0]), t(pa[1]), t(pa[2]), t(pa[3])
    DL, DR = torch.empty_like(SL), torch.empty_like(SR)
    with mk(pm, 0, patch=5, iters=2, rows=rows, cols=cols) as e:
        run = lambda r_=rows, c_=cols: e.match_device(1, L.data_ptr(), R.data_ptr(), r_, c_, SL.data_ptr(),
                                                        SR.data_ptr(), DL.data_ptr(), DR.data_ptr())
        # 1) capturing a size that was never matched: the noise table cannot be built inside a capture
        e.capture_begin()
        with pytest.raises(pm.PmError) as ei:
            run()
        assert ei.value.status == pm.PM_ERR_BUSY
        with pytest.raises(pm.PmError):
            e.capture_end()  # the failed call already ended the capture
        run()                # ... and the handle still works
        e.synchronize()
        el, er = oracle.match(oparams(oracle, 0, 5, 2), *pa[:4])
        assert_same(DL.cpu().numpy(), el, "after an aborted capture")
        # 2) refused while capturing
        e.capture_begin()
        for call in (e.synchronize, lambda: e.profile_enable(True), e.profile_read,
                     lambda: e.match(pa[0], pa[1], pa[2], pa[3]), lambda: e.gradient_magnitude(pa[0]), e.capture_begin):
            with pytest.raises(pm.PmError) as ei:
                call()
            assert ei.value.status == pm.PM_ERR_BUSY
        run()
        e.capture_end()
        e.replay()
        e.synchronize()
        assert_same(DL.cpu().numpy(), el, "replay after refused calls")
        # 3) an oversized request inside a capture fails cleanly too
        e.capture_begin()
        with pytest.raises(pm.PmError) as ei:
            run(rows + 8, cols)
        assert ei.value.status == pm.PM_ERR_SIZE
        run()
        e.synchronize()
    # 4) A capture whose recorded calls forked work onto another stream and never joined it back: ending such a capture
    # faults inside the runtime (gpurun_out/r03/crash.log) -- pm_capture_end joins, discards and says PM_ERR_STATE.
    with mk(pm, 0, patch=5, iters=2, rows=rows, cols=cols) as e:
        run2 = lambda: e.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(),
                                      DL.data_ptr(), DR.data_ptr())
        with pytest.raises(pm.PmError) as ei:
            e.debug_capture_fork()   # not capturing
        assert ei.value.status == pm.PM_ERR_STATE
        run2()
        e.synchronize()
        e.capture_begin()
        run2()
        e.debug_capture_fork()
        with pytest.raises(pm.PmError) as ei:
            e.capture_end()
        assert ei.value.status == pm.PM_ERR_STATE and "never joined" in str(ei.value)
        with pytest.raises(pm.PmError):
            e.replay()               # nothing was kept
        DL.zero_()
        run2()                       # the handle works, and captures, as before
        e.synchronize()
        assert_same(DL.cpu().numpy(), el, "after a discarded capture")
        e.capture_begin()
        run2()
        e.capture_end()
        DL.zero_()
        e.replay()
        e.synchronize()
        assert_same(DL.cpu().numpy(), el, "capture after a discarded capture")
    # destroying a handle in the middle of a capture must not hang or crash
    e2 = mk(pm, 0, patch=5, iters=1, rows=rows, cols=cols)
    e2.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(),
                    DR.data_ptr())
    e2.synchronize()
    e2.capture_begin()
    e2.close()


@pytest.mark.parametrize("sem", SEMS)
def test_one_view_device_match_with_caller_gradients(pm, oracle, synth, sem):
    """PatchmatchGpu::Match(GpuMat iml, imr, Gl, Gr, GpuMat& disp) (patchmatch_gpu.h:104-108): float images and
    gradients with a row step, disp = seed in / result out, enqueued behind the caller's stream."""
    torch = pytest.importorskip("torch")
    rows, cols, padc = 64, 96, 104
    dev = torch.device("cuda:0")
    l, r, sl, sr, _ = small_pair(synth, 94 + sem, rows, cols, n_points=25, dilate_factor=2)

    def padded(a):
        buf = torch.full((rows, padc), -7.0, dtype=torch.float32, device=dev)
        buf[:, :cols] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        return buf

    with mk(pm, sem, patch=5, iters=3, rows=rows, cols=cols) as e:
        gl, gr = e.gradient_magnitude(l), e.gradient_magnitude(r)
        IL, IR, GL, GR, D = padded(l), padded(r), padded(gl), padded(gr), padded(sl)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            D2 = D.clone()  # produced on the caller's stream: the call must wait for it
            e.match_view_device(IL.data_ptr(), IR.data_ptr(), GL.data_ptr(), GR.data_ptr(), rows, cols, padc * 4,
                                D2.data_ptr(), padc * 4, stream=side.cuda_stream)
            out = D2.clone()  # consumed on the caller's stream: must see the result
        side.synchronize()
        e.match_view_device(IL.data_ptr(), IR.data_ptr(), GL.data_ptr(), GR.data_ptr(), rows, cols, padc * 4,
                            D.data_ptr(), padc * 4)
        e.synchronize()
        with pytest.raises(pm.PmError):
            e.match_view_device(IL.data_ptr(), IR.data_ptr(), GL.data_ptr(), GR.data_ptr(), rows, cols, cols * 4 - 4,
                                D.data_ptr())
    want = oracle.match(oparams(oracle, sem, 5, 3, lr=0), l, r, sl, None)[0]
    assert_same(D[:, :cols].cpu().numpy(), want, "one view, own stream")
    assert_same(out[:, :cols].cpu().numpy(), want, "one view, caller's stream")
    assert (D[:, cols:] == -7.0).all()  # the padding behind a row is not touched


def test_set_unit_noise_and_seed_rules(pm, oracle, synth):
    rows, cols = 48, 80
    l, r, sl, sr, _ = small_pair(synth, 96, rows, cols, n_points=20, dilate_factor=2)
    with mk(pm, 1, iters=3, rows=rows, cols=cols) as e:
        base = e.match(l, r, sl, sr)
        e.set_unit_noise(e.unit_noise(rows, cols))  # the table the engine already uses: nothing changes
        again = e.match(l, r, sl, sr)
        assert_same(again[0], base[0], "same table")
        e.set_unit_noise(np.zeros((rows, cols), np.float32))  # no noise at all == amplitude 0 in the oracle
        quiet = e.match(l, r, sl, sr)
    want = oracle.match(oparams(oracle, 1, 3, 3, noise_amp=[0.0] * 16), l, r, sl, sr)
    assert_same(quiet[0], want[0], "zero noise table, left")
    assert_same(quiet[1], want[1], "zero noise table, right")
    # with sparse_init a batch gives a view's seed maps for all pairs or for none
    with mk(pm, 1, iters=1, rows=rows, cols=cols, batch=2, sparse_init=1) as e:
        with pytest.raises(pm.PmError) as ei:
            e.match_batch([l, l], [r, r], [sl, None], None)
        assert ei.value.status == pm.PM_ERR_INVALID_ARG
    # the retired engines are refused
    for eng in (3, 4, 6):
        with pytest.raises(pm.PmError):
            mk(pm, 0, engine=eng)


def test_random_configurations_match_the_oracle(pm, oracle, synth):
    """Property test (hypothesis, fixed seed): random small sizes, windows, iteration counts, noise schedules,
    group widths and both semantics -- the default engine equals the oracle bit for bit."""
    hyp = pytest.importorskip("hypothesis")
    st = pytest.importorskip("hypothesis.strategies")

    @hyp.settings(max_examples=80, deadline=None, derandomize=True, database=None,
                  suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(rows=st.integers(8, 44), cols=st.integers(8, 90), sem=st.sampled_from([0, 1]),
               patch=st.sampled_from([3, 5, 7, 9, 11]), iters=st.integers(0, 4),
               amp0=st.sampled_from([32.0, 6.0, 1.0, 0.25]), lr=st.booleans(), seed=st.integers(0, 10 ** 6),
               density=st.sampled_from([0, 3, 25]))
    def run(rows, cols, sem, patch, iters, amp0, lr, seed, density):
        hyp.assume(rows > patch + 2 and cols > patch + 2)
        rng = np.random.default_rng(seed)
        d_true = int(rng.integers(0, max(1, min(12, cols // 4))))
        base = rng.integers(0, 256, (rows, cols + d_true), dtype=np.uint8)
        base = (base.astype(np.float32) * 0.5 + np.roll(base, 1, 1) * 0.5).astype(np.uint8)  # some structure
        left = np.ascontiguousarray(base[:, d_true:d_true + cols])
        right = np.ascontiguousarray(base[:, :cols])
        seed_l = np.zeros((rows, cols), np.float32)
        if density:
            ys, xs = rng.integers(0, rows, density), rng.integers(0, cols, density)
            seed_l[ys, xs] = rng.uniform(0.5, max(1.0, d_true + 2.0), density).astype(np.float32)
        seed_r = np.ascontiguousarray(seed_l[:, ::-1]) if lr else None
        params = pm.default_params(sem, patch=patch, patchmatch_iters=iters, left_right_check=1 if lr else 0)
        op = oparams(oracle, sem, patch, iters)
        op.left_right_check = 1 if lr else 0
        for i in range(iters):
            params.noise_amp[i] = amp0 / (2 ** i)
            op.noise_amp[i] = amp0 / (2 ** i)
        with pm.Engine(params, max_rows=rows, max_cols=cols) as e:
            dl, dr = e.match(left, right, seed_l, seed_r)
        el, er = oracle.match(op, left, right, seed_l, seed_r)
        assert_same(dl, el, f"left {rows}x{cols} sem{sem} p{patch} it{iters} amp{amp0} lr{lr}")
        if lr:
            assert_same(dr, er, "right")

    run()


# ---- full size: properties and engine-vs-engine ------------------------------------------------------------
def test_full_size_baseline_config_properties(pm, oracle, synth):
    """BASELINE.json configs[1]: one 1280x720 pair, 8 iterations, 11x11 window."""
    rows, cols, patch, iters = 720, 1280, 11, 8
    p = synth.make_pair(0, rows, cols)
    l, r, sl, sr = p["left"], p["right"], p["seed_l"], p["seed_r"]
    with mk(pm, 0, 0, patch=patch, iters=iters, rows=rows, cols=cols) as e:   # the engine bench.py runs
        dl, dr = e.match(l, r, sl, sr)
        dl2, dr2 = e.match(l, r, sl, sr)
    with mk(pm, 0, 1, patch=patch, iters=iters, rows=rows, cols=cols) as e:
        sl_, sr_ = e.match(l, r, sl, sr)
    with mk(pm, 0, 2, patch=patch, iters=iters, rows=rows, cols=cols) as e:
        wl_, wr_ = e.match(l, r, sl, sr)
    assert_same(dl, dl2, "run-to-run determinism (left)")
    assert_same(dr, dr2, "run-to-run determinism (right)")
    assert_same(dl, sl_, "default engine == serial anchor (left)")
    assert_same(dr, sr_, "default engine == serial anchor (right)")
    assert_same(wl_, sl_, "wave engine == serial anchor (left)")
    assert_same(wr_, sr_, "wave engine == serial anchor (right)")
    assert np.isfinite(dl).all() and np.isfinite(dr).all() and dl.min() >= 0 and dr.min() >= 0
    h = patch // 2
    xs = np.arange(cols, dtype=np.float32)[None, :]
    assert np.all(dl[h:rows - h, h:cols - h] <= (xs - h)[:, h:cols - h])          # patchmatch.cpp:175
    assert np.all(dr[h:rows - h, h:cols - h] <= (cols - 1 - xs - h)[:, h:cols - h])
    # border pixels are never swept (patchmatch.cpp:267-270): they only see the noise passes
    unit = oracle.rng_fill_uniform(rows * cols, -1.0, 1.0).reshape(rows, cols)
    border = sl.copy()
    for i in range(iters):
        border = oracle.gpu_add_foreground_noise(border, unit, 32.0 / 2 ** i)
    bl = oracle.gpu_mask_occlusions(border, dr)  # the cross-check also applies to border pixels
    for sl_b in (np.s_[:h, :], np.s_[rows - h:, :], np.s_[:, :h], np.s_[:, cols - h:]):
        # left border pixels: noise-only value, then MaskOcclusions against the final right map
        assert_same(dl[sl_b], bl[sl_b], "untouched border")
    # MaskOcclusions is idempotent on its own output
    assert_same(oracle.gpu_mask_occlusions(dl, dr), dl, "cross-check idempotence")
    # and the result is a disparity map: most foreground pixels within 1 px of the synthetic truth
    fg = dl > 0
    assert fg.mean() > 0.15 and (np.abs(dl - p["gt"])[fg] < 1.0).mean() > 0.95
    # a 64-row band of the same pair is an independent problem the oracle can afford at 11x11 / 8 it.
    band = np.s_[300:364, :]
    with mk(pm, 0, 0, patch=patch, iters=iters, rows=64, cols=cols) as e:
        bdl, bdr = e.match(l[band], r[band], sl[band], sr[band])
    el, er = oracle.match(oparams(oracle, 0, patch, iters), l[band], r[band], sl[band], sr[band])
    assert_same(bdl, el, "64-row band at full width, left")
    assert_same(bdr, er, "64-row band at full width, right")


def test_full_size_headline_whole_frame_equals_oracle(pm, oracle, synth):
    """BASELINE.json configs[1], the benchmarked configuration, WHOLE frame against the oracle: 1280x720, 8 iterations,
    11x11, PM_SEM_CPU, both views + cross-check, tolerance 0.  The oracle runs its literal form (getRectSubPix patches +
    the test's functor, patchmatch.cpp:158-196 / patchmatch_test.cpp:30-45) on all host cores: rows / columns of a sweep
    are independent chains, so the thread count does not change the result (test_oracle_algorithm.py).  The reference
    pattern: the whole-image recipe of patchmatch_test.cpp:149-183."""
    import os
    rows, cols, patch, iters = 720, 1280, 11, 8
    p = synth.make_pair(0, rows, cols)
    l, r, sl, sr = p["left"], p["right"], p["seed_l"], p["seed_r"]
    with mk(pm, 0, 0, patch=patch, iters=iters, rows=rows, cols=cols) as e:   # the engine bench.py runs
        dl, dr = e.match(l, r, sl, sr)
    op = oracle.default_params(0, patch=patch, n_iters=iters, left_right_check=1, literal=1,
                               nthreads=min(os.cpu_count() or 1, 16))
    el, er = oracle.match(op, l, r, sl, sr)
    assert_same(dl, el, "whole 1280x720 frame, left")
    assert_same(dr, er, "whole 1280x720 frame, right")


def test_full_size_batch_of_32_pairs(pm, oracle, synth):
    """BASELINE configs[2]'s per-GPU share: 32 slots of 1280x720 (8 iterations, 11x11) in ONE pm_match_device call.
    Every slot of the default engine must equal the serial anchor's map of its pair, and a 64-row band of one of the
    pairs must equal the oracle."""
    torch = pytest.importorskip("torch")
    rows, cols, patch, iters, nb = 720, 1280, 11, 8, 32
    dev = torch.device("cuda:0")
    uniq = [synth.make_pair(10 + i, rows, cols) for i in range(4)]
    slot_pair = [(3 * i + i // 4) % 4 for i in range(nb)]  # an irregular assignment of the 4 pairs to the 32 slots
    st = lambda k, idx: torch.from_numpy(np.stack([uniq[j][k] for j in idx])).to(dev).contiguous()
    L, R, SL, SR = (st(k, slot_pair) for k in ("left", "right", "seed_l", "seed_r"))
    DL = torch.empty((nb, rows, cols), dtype=torch.float32, device=dev)
    DR = torch.empty_like(DL)
    with mk(pm, 0, 0, patch=patch, iters=iters, rows=rows, cols=cols, batch=nb) as e:
        e.match_device(nb, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(),
                       DR.data_ptr())
        e.synchronize()
    L4, R4, SL4, SR4 = (st(k, range(4)) for k in ("left", "right", "seed_l", "seed_r"))
    AL = torch.empty((4, rows, cols), dtype=torch.float32, device=dev)
    AR = torch.empty_like(AL)
    with mk(pm, 0, 1, patch=patch, iters=iters, rows=rows, cols=cols, batch=4) as e:  # PM_ENGINE_SERIAL
        e.match_device(4, L4.data_ptr(), R4.data_ptr(), rows, cols, SL4.data_ptr(), SR4.data_ptr(), AL.data_ptr(),
                       AR.data_ptr())
        e.synchronize()
    for i in range(nb):
        assert torch.equal(DL[i], AL[slot_pair[i]]), f"slot {i} (pair {slot_pair[i]}), left"
        assert torch.equal(DR[i], AR[slot_pair[i]]), f"slot {i} (pair {slot_pair[i]}), right"
    p = uniq[1]
    band = np.s_[200:264, :]
    with mk(pm, 0, 0, patch=patch, iters=iters, rows=64, cols=cols) as e:
        bdl, bdr = e.match(p["left"][band], p["right"][band], p["seed_l"][band], p["seed_r"][band])
    el, er = oracle.match(oparams(oracle, 0, patch, iters), p["left"][band], p["right"][band], p["seed_l"][band],
                          p["seed_r"][band])
    assert_same(bdl, el, "64-row band of pair 11, left")
    assert_same(bdr, er, "64-row band of pair 11, right")


@pytest.mark.parametrize("engine", [2, 5])
def test_full_size_gpu_semantics_engines_agree(pm, oracle, synth, engine):
    rows, cols = 720, 1280
    p = synth.make_pair(1, rows, cols)
    with mk(pm, 1, engine, iters=3, rows=rows, cols=cols) as e:
        dl, dr = e.match(p["left"], p["right"], p["seed_l"], p["seed_r"])
    el, er = oracle.match(oparams(oracle, 1, 3, 3), p["left"], p["right"], p["seed_l"], p["seed_r"])
    assert_same(dl, el, "1280x720 PM_SEM_GPU left vs oracle")
    assert_same(dr, er, "1280x720 PM_SEM_GPU right vs oracle")


@pytest.mark.gpu
def test_plan_beyond_the_32_bit_plane_offsets_is_refused(pm):
    """The sweep kernels address a view's planes with 32-bit byte offsets: pm_create refuses such a plan before any
    allocation."""
    with pytest.raises(pm.PmError) as ex:
        pm.Engine(pm.default_params(0), max_rows=16384, max_cols=16384)
    assert ex.value.status == pm.PM_ERR_SIZE


@pytest.mark.gpu
def test_differential_fuzz_product_engine_vs_serial_anchor():
    """tools/fuzz_engines.py: random sizes / windows / iteration counts / noise schedules / seed fields (incl. sample
    positions at the lerp-weight extremes), whole Match() of the product engine == the serial anchor, bit for bit.
    (4500 cases were run when the packed-SAD / dot2 / record-plane kernels went in; 60 here.)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_engines.py"), "--cases", "60", "--seed", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_differential_fuzz_of_the_entry_points():
    """tools/fuzz_api.py: batches, the submit / collect queue at random depths, device-resident batches, strided host
    buffers, sizes below the plan, captured graphs replayed on new data -- every entry point == pm_match_u8 of the
    same parameters, both scalar semantics and the plane mode (4000 cases were run; 60 here)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_api.py"), "--cases", "60", "--seed", "6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]

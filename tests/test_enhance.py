"""Row f-2 of SURVEY.md section 8: the range-free "stereo-ready" enhancement in front of stereo
(pm_stereo_ready / pm_gaussian_blur / pm_normalize, include/pm/imaging.h).

Oracle: oracle/pm_enhance_oracle.c (parity unpinned: OpenCV primitives restated from memory, no reference
outputs exist).  Device results are compared with the oracle bit for bit -- both sides perform the same single
IEEE operations in the same order."""
import numpy as np
import pytest

import oracle_lib as O


def color_image(rows, cols, seed):
    """A smooth underwater-like colour cast plus texture, 8-bit BGR."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    base = np.stack([0.55 + 0.25 * np.sin(xx / cols * 3.0), 0.45 + 0.2 * np.cos(yy / rows * 2.0),
                     0.15 + 0.1 * np.sin((xx + yy) / (rows + cols) * 4.0)], -1)
    tex = rng.uniform(-0.12, 0.12, (rows, cols, 1)).astype(np.float32) * np.array([1.0, 0.8, 0.5], np.float32)
    return np.clip((base + tex) * 255.0, 0, 255).astype(np.uint8)


# ---- oracle known answers (CPU) --------------------------------------------------------------------------
def test_oracle_gaussian_kernel_and_blur_properties():
    k = O.gaussian_kernel(33, 33 / 4.0)
    assert k.dtype == np.float32 and abs(float(k.astype(np.float64).sum()) - 1.0) < 1e-6
    assert np.array_equal(k, k[::-1]) and k.argmax() == 16
    # closed form of cv::getGaussianKernel
    x = np.arange(33) - 16.0
    t = np.exp(-0.5 * x * x / (8.25 * 8.25)).astype(np.float32)
    assert np.array_equal(k, (t * (1.0 / t.astype(np.float64).sum())).astype(np.float32))
    # a constant image stays constant (up to the rounding of the tap sum); replicate border: no darkening
    img = np.full((20, 30, 3), 0.37, np.float32)
    out = O.gaussian_blur(img, 9, 2.25)
    assert np.allclose(out, 0.37, rtol=0, atol=2e-7) and abs(out[0, 0, 0] - out[10, 15, 0]) < 1e-7
    # an impulse reproduces the outer product of the taps
    img = np.zeros((21, 21), np.float32)
    img[10, 10] = 1.0
    k9 = O.gaussian_kernel(9, 2.25)
    out = O.gaussian_blur(img, 9, 2.25)
    assert np.allclose(out[6:15, 6:15], np.outer(k9, k9), rtol=1e-6, atol=0)


def test_oracle_normalize_stretches_value_and_keeps_hue():
    img = color_image(64, 96, 1).astype(np.float32) / 255.0
    out = O.normalize(img)
    V_in, V_out = img.max(-1), out.max(-1)
    lo, hi = O.value_minmax_eighth(V_in)
    assert 0.0 < lo < hi <= 1.0
    # V' = (V - lo) / (hi - lo): checked through the max channel (lo comes from the smoothed 1/8 image, so a
    # few pixels fall below it and get a negative value, whose largest channel is no longer V')
    ok = V_in > lo
    assert ok.mean() > 0.9
    assert np.allclose(V_out[ok], ((V_in - lo) / (hi - lo))[ok], rtol=2e-5, atol=2e-6)
    # hue and saturation are untouched: channel ratios relative to the value are preserved
    far = V_in > lo + 0.02
    assert np.allclose((out / V_out[..., None])[far], (img / V_in[..., None])[far], rtol=0, atol=2e-4)
    # a gray image (s = 0) only gets stretched
    gray = np.repeat(np.linspace(0.2, 0.8, 64 * 96, dtype=np.float32).reshape(64, 96, 1), 3, -1)
    og = O.normalize(gray)
    assert np.array_equal(og[..., 0], og[..., 1]) and np.array_equal(og[..., 1], og[..., 2])


def test_oracle_stereo_ready_flattens_the_colour_cast():
    bgr8 = color_image(72, 120, 2)
    J, gray = O.stereo_ready(bgr8)
    assert J.shape == (72, 120, 3) and gray.shape == (72, 120) and gray.dtype == np.uint8
    # illuminant division: the strong blue / weak red cast is gone (channel means within 25 % of each other)
    m = J.reshape(-1, 3).mean(0)
    assert m.max() / m.min() < 1.25, m
    src = bgr8.reshape(-1, 3).astype(np.float32).mean(0)
    assert src.max() / src.min() > 2.0
    g = J[..., 0] * np.float32(0.114) + J[..., 1] * np.float32(0.587) + J[..., 2] * np.float32(0.299)
    assert np.abs(gray.astype(np.float32) - np.clip(np.rint(g * 255), 0, 255)).max() <= 1


def test_oracle_against_independent_implementations():
    """scipy's separable correlation (replicate border) and the standard library's colorsys agree with the
    oracle's Gaussian blur and HSV round trip to float accuracy (different summation order / formulas)."""
    import colorsys
    from scipy import ndimage
    rng = np.random.default_rng(8)
    img = rng.uniform(0, 1, (40, 56, 3)).astype(np.float32)
    k = O.gaussian_kernel(21, 21 / 4.0).astype(np.float64)
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="nearest"), k, axis=0,
                              mode="nearest")
    assert np.allclose(O.gaussian_blur(img, 21, 21 / 4.0), ref, rtol=0, atol=2e-6)
    # Normalize with a known stretch: V' = (V - lo) / (hi - lo), hue and saturation kept (colorsys)
    out = O.normalize(img)
    lo, hi = O.value_minmax_eighth(img.max(-1))
    for (y, x) in [(0, 0), (7, 13), (20, 41), (39, 55), (11, 2)]:
        b, g, r = (float(c) for c in img[y, x])
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        rr, gg, bb = colorsys.hsv_to_rgb(h, s, (v - lo) / (hi - lo))
        assert np.allclose(out[y, x], (bb, gg, rr), rtol=0, atol=3e-5), (y, x)
    # the 1/8 bilinear resize is an average of 4 neighbours at the centre of each 8x8 cell
    V = rng.uniform(0, 1, (32, 48)).astype(np.float32)
    cells = np.stack([V[8 * j + 3:8 * j + 5, 8 * i + 3:8 * i + 5].mean() for j in range(4) for i in range(6)])
    lo, hi = O.value_minmax_eighth(V)
    assert abs(lo - cells.min()) < 1e-6 and abs(hi - cells.max()) < 1e-6


# ---- device parity ------------------------------------------------------------------------------------------
def _dev(t, a):
    return t.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,ch,ksize", [(40, 64, 3, 21), (33, 300, 1, 99), (70, 50, 3, 17), (9, 700, 3, 233),
                                                (300, 40, 2, 13)])
def test_device_gaussian_blur_is_bit_exact(pm, rows, cols, ch, ksize):
    import torch
    rng = np.random.default_rng(rows + cols)
    img = rng.uniform(0, 1, (rows, cols, ch) if ch > 1 else (rows, cols)).astype(np.float32)
    sigma = ksize / 4.0
    with pm.Engine(pm.default_params(0, patch=5), max_rows=16, max_cols=16) as e:
        d_in = _dev(torch, img)
        d_out = torch.empty_like(d_in)
        e.gaussian_blur(d_in.data_ptr(), rows, cols, ch, ksize, sigma, d_out.data_ptr())
        e.synchronize()
    want = O.gaussian_blur(img, ksize, sigma)
    assert np.array_equal(d_out.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(64, 96), (50, 77), (16, 24), (120, 333)])
def test_device_normalize_and_stereo_ready_are_bit_exact(pm, rows, cols):
    import torch
    bgr8 = color_image(rows, cols, rows * 3 + cols)
    with pm.Engine(pm.default_params(0, patch=5), max_rows=16, max_cols=16) as e:
        f = bgr8.astype(np.float32) * np.float32(1.0 / 255.0)
        d_f = _dev(torch, f)
        d_n = torch.empty_like(d_f)
        e.normalize(d_f.data_ptr(), rows, cols, d_n.data_ptr())
        e.synchronize()
        assert np.array_equal(d_n.cpu().numpy(), O.normalize(f)), "Normalize"

        d_b = _dev(torch, bgr8)
        d_J = torch.empty((rows, cols, 3), device="cuda")
        d_g = torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
        e.stereo_ready(d_b.data_ptr(), rows, cols, d_J.data_ptr(), d_g.data_ptr())
        e.synchronize()
        J, gray = O.stereo_ready(bgr8)
        assert np.array_equal(d_J.cpu().numpy(), J), "J = Normalize(NormalizeColorIlluminant(I))"
        assert np.array_equal(d_g.cpu().numpy(), gray), "8-bit gray"
        # gray-only call gives the same image
        d_g2 = torch.zeros_like(d_g)
        e.stereo_ready(d_b.data_ptr(), rows, cols, None, d_g2.data_ptr())
        e.synchronize()
        assert torch.equal(d_g, d_g2)
        with pytest.raises(pm.PmError):
            e.stereo_ready(d_b.data_ptr(), 4, 4, None, d_g2.data_ptr())


@pytest.mark.gpu
def test_enhanced_pair_feeds_match_without_host_round_trip(pm, oracle, synth):
    """config 5's plumbing: colour pair -> stereo-ready gray on the device -> Match() on the same stream."""
    import torch
    rows, cols = 96, 160
    p = synth.make_pair(9, rows, cols)
    # colourise the synthetic pair with a depth-independent cast (the enhancement is range-free)
    cast = np.array([0.9, 0.7, 0.35], np.float32)
    lc = np.clip(p["left"][..., None].astype(np.float32) * cast, 0, 255).astype(np.uint8)
    rc = np.clip(p["right"][..., None].astype(np.float32) * cast, 0, 255).astype(np.uint8)
    with pm.Engine(pm.default_params(0, patch=5, patchmatch_iters=2), max_rows=rows, max_cols=cols) as e:
        d_l, d_r = _dev(torch, lc), _dev(torch, rc)
        g_l = torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
        g_r = torch.empty_like(g_l)
        SL, SR = _dev(torch, p["seed_l"]), _dev(torch, p["seed_r"])
        DL, DR = torch.empty_like(SL), torch.empty_like(SR)
        e.stereo_ready(d_l.data_ptr(), rows, cols, None, g_l.data_ptr())
        e.stereo_ready(d_r.data_ptr(), rows, cols, None, g_r.data_ptr())
        e.match_device(1, g_l.data_ptr(), g_r.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(),
                       DR.data_ptr())
        e.synchronize()
    _, gl = O.stereo_ready(lc)
    _, gr = O.stereo_ready(rc)
    el, er = oracle.match(oracle.default_params(0, patch=5, n_iters=2, nthreads=8), gl, gr, p["seed_l"], p["seed_r"])
    assert np.array_equal(DL.cpu().numpy(), el) and np.array_equal(DR.cpu().numpy(), er)
    assert (el > 0).mean() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["scalar", "planes_f16", "planes_f32"])
def test_fused_bgr_match_equals_stereo_ready_then_match(pm, synth, mode):
    """pm_match_bgr_device (the enhancement's per-pixel tail inside the prep kernel, no gray image in memory) gives the
    maps of pm_stereo_ready x 2 + pm_match_device bit for bit -- scalar mode and the plane mode with both state types,
    a batch of two pairs, an image size that is not a multiple of the prep tile."""
    import torch
    rows, cols, n = 101, 163, 2
    pairs = [synth.make_pair(20 + i, rows, cols) for i in range(n)]
    bl = np.stack([synth.to_bgr(p["left"], 3 + i) for i, p in enumerate(pairs)])
    br = np.stack([synth.to_bgr(p["right"], 7 + i) for i, p in enumerate(pairs)])
    if mode == "scalar":
        prm = pm.default_params(0, patch=7, patchmatch_iters=2, sparse_init=1)
    else:
        prm = pm.default_params(0, patch=7, patchmatch_iters=2, mode=pm.PM_MODE_PLANES, max_disp=64,
                                state_dtype=pm.PM_STATE_F16 if mode == "planes_f16" else pm.PM_STATE_F32)
    dev = torch.device("cuda")
    BL, BR = torch.from_numpy(bl).to(dev).contiguous(), torch.from_numpy(br).to(dev).contiguous()
    GL = torch.empty((n, rows, cols), dtype=torch.uint8, device=dev)
    GR = torch.empty_like(GL)
    D = [torch.empty((n, rows, cols), dtype=torch.float32, device=dev) for _ in range(4)]
    px = rows * cols
    with pm.Engine(prm, max_rows=rows, max_cols=cols, max_batch=n) as e:
        for i in range(n):
            e.stereo_ready(BL.data_ptr() + 3 * px * i, rows, cols, None, GL.data_ptr() + px * i)
            e.stereo_ready(BR.data_ptr() + 3 * px * i, rows, cols, None, GR.data_ptr() + px * i)
        e.match_device(n, GL.data_ptr(), GR.data_ptr(), rows, cols, None, None, D[0].data_ptr(), D[1].data_ptr())
        e.match_bgr_device(n, BL.data_ptr(), BR.data_ptr(), rows, cols, None, None, D[2].data_ptr(), D[3].data_ptr())
        e.synchronize()
    assert torch.equal(D[0], D[2]) and torch.equal(D[1], D[3])
    assert float((D[2] > 0).float().mean()) > 0.2

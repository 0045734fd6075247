"""Third-party cross-check of the oracle's OpenCV primitives against scipy.ndimage (CPU only).

The oracle restates OpenCV 3.4 primitives from the published algorithms (oracle/pm_oracle.c, pm_seed_oracle.c,
pm_enhance_oracle.c); OpenCV itself is not in the image and the reference's tests pin nothing, so parity stays
"unpinned" (DESIGN.md 2).  What this file narrows: the FILTERING conventions -- kernel taps, signs, border rules,
interpolation geometry -- are checked against scipy's independent implementations of the same textbook operators, so an
error there would have to be shared with scipy.  What it cannot check: OpenCV's own rounding / fixed-point choices
(the 8-bit getRectSubPix weights, cv::RNG, INTER_LINEAR's 11-bit coefficients), which stay this build's definitions.
Call sites in the reference: cv::Sobel (test/stereo_matching/patchmatch_test.cpp:48-64), cv::dilate
(src/vehicle/stereo_matching/patchmatch.cpp:76-79), cv::getRectSubPix (patchmatch.cpp:101-109), cv::GaussianBlur
(src/vehicle/imaging/illuminant.cpp:10-21), cv::goodFeaturesToTrack (feature_tracking/feature_detector.cpp:44-57).
"""
import numpy as np
import pytest

ndi = pytest.importorskip("scipy.ndimage")


@pytest.mark.parametrize("shape", [(5, 7), (16, 16), (33, 20), (2, 9), (64, 48)])
def test_sobel_magnitude_equals_scipy_sobel_with_mirror_border(oracle, shape):
    """cv::Sobel(ksize 3), BORDER_REFLECT_101 == scipy.ndimage.sobel(mode="mirror"): derivative [-1 0 1] along the axis,
    smoothing [1 2 1] across, unnormalised.  Both derivative images are small integers, so the float32 magnitude
    sqrt(sx^2 + sy^2) (exact sum below 2^24, correctly rounded sqrt) must agree BIT FOR BIT."""
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    f = im.astype(np.float64)
    sx = ndi.sobel(f, axis=1, mode="mirror")
    sy = ndi.sobel(f, axis=0, mode="mirror")
    want = np.sqrt((sx * sx + sy * sy).astype(np.float32))
    assert np.array_equal(oracle.gradient_magnitude(im), want.astype(np.float32))


@pytest.mark.parametrize("k", [1, 2, 3, 17])
def test_rect_dilate_equals_scipy_grey_dilation(oracle, k):
    """cv::dilate with a (2k+1)^2 rectangle, anchor at the centre, border samples ignored (patchmatch.cpp:76-79) ==
    grey_dilation with a constant -inf border."""
    rng = np.random.default_rng(k)
    src = np.where(rng.random((40, 57)) < 0.03, rng.uniform(1, 90, (40, 57)), 0).astype(np.float32)
    want = ndi.grey_dilation(src, size=(2 * k + 1, 2 * k + 1), mode="constant", cval=-np.inf)
    assert np.array_equal(oracle.dilate_rect(src, k), want)


def _patch_coords(pw, ph, cx, cy):
    # cv::getRectSubPix: the patch's top-left sample sits at centre - (size - 1) / 2
    ys = cy - (ph - 1) * 0.5 + np.arange(ph)
    xs = cx - (pw - 1) * 0.5 + np.arange(pw)
    return np.meshgrid(ys, xs, indexing="ij")


@pytest.mark.parametrize("cx,cy", [(10.0, 9.0), (10.25, 9.5), (3.7, 2.1), (0.4, 0.2), (30.9, 21.6), (-1.5, 12.0)])
def test_rect_subpix_f32_equals_scipy_bilinear_with_replicated_border(oracle, cx, cy):
    """The 32f -> 32f path (gradient patches, patchmatch.cpp:109): bilinear interpolation, replicate border ==
    map_coordinates(order=1, mode="nearest").  scipy interpolates in binary64, OpenCV in binary32 with four
    pre-multiplied weights: agreement to 1e-4 relative (values reach ~1400)."""
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 1400, (24, 33)).astype(np.float32)
    for pw, ph in ((3, 3), (7, 5), (11, 11)):
        yy, xx = _patch_coords(pw, ph, cx, cy)
        want = ndi.map_coordinates(img.astype(np.float64), [yy, xx], order=1, mode="nearest")
        got = oracle.get_rect_subpix(img, pw, ph, cx, cy)
        assert np.allclose(got, want, rtol=1e-4, atol=1e-3), (pw, ph, np.abs(got - want).max())


@pytest.mark.parametrize("cx,cy", [(10.0, 9.0), (10.25, 9.5), (3.7, 2.1), (0.4, 0.2), (30.9, 21.6)])
def test_rect_subpix_u8_is_the_rounded_scipy_bilinear(oracle, cx, cy):
    """The 8u -> 8u path (image patches, patchmatch.cpp:101): the same geometry with fixed-point weights and
    round-to-nearest -- within one grey level of the exact bilinear value everywhere, equal to its rounding wherever
    the exact value is not within 2^-7 of a tie."""
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (24, 33), dtype=np.uint8)
    for pw, ph in ((3, 3), (7, 5), (11, 11)):
        yy, xx = _patch_coords(pw, ph, cx, cy)
        exact = ndi.map_coordinates(img.astype(np.float64), [yy, xx], order=1, mode="nearest")
        got = oracle.get_rect_subpix(img, pw, ph, cx, cy).astype(np.float64)
        assert np.abs(got - exact).max() <= 0.5 + 2.0 ** -7
        clear = np.abs((exact - np.floor(exact)) - 0.5) > 2.0 ** -7
        assert np.array_equal(got[clear], np.rint(exact[clear]))


@pytest.mark.parametrize("n,sigma", [(5, 1.0), (11, 2.5), (61, 10.0), (427, 71.0)])
def test_gaussian_kernel_equals_scipy_weights(oracle, n, sigma):
    """cv::getGaussianKernel(n, sigma > 0): exp(-(i - c)^2 / (2 sigma^2)) normalised to sum 1 == the weights
    scipy.ndimage.gaussian_filter1d uses at radius (n - 1) / 2 (427 taps = the illuminant blur of a 1280-wide image)."""
    r = (n - 1) // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    want = np.exp(-0.5 * x * x / (sigma * sigma))
    want /= want.sum()
    impulse = np.zeros(2 * n + 1)
    impulse[n] = 1.0
    scipy_w = ndi.gaussian_filter1d(impulse, sigma, mode="constant", radius=r)[n - r:n + r + 1]
    assert np.allclose(scipy_w, want, rtol=1e-12)                      # scipy's weights ARE that formula
    assert np.allclose(oracle.gaussian_kernel(n, sigma), want, rtol=2e-6)  # ... and so are the oracle's (float32)


def test_gaussian_blur_equals_scipy_filter_with_replicated_border(oracle):
    """cv::GaussianBlur(..., cv::BORDER_REPLICATE) as the illuminant estimate calls it (imaging/illuminant.cpp:16) == two
    gaussian_filter1d passes with mode="nearest" at the same radius; float32 accumulation against float64: 1e-4."""
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 255, (40, 56)).astype(np.float32)
    n, sigma = 21, 3.5
    r = (n - 1) // 2
    want = ndi.gaussian_filter1d(ndi.gaussian_filter1d(img.astype(np.float64), sigma, axis=1, mode="nearest", radius=r),
                                 sigma, axis=0, mode="nearest", radius=r)
    got = oracle.gaussian_blur(img, n, sigma)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-3), np.abs(got - want).max()


@pytest.mark.parametrize("block", [3, 5, 7])
def test_min_eigenvalue_map_from_scipy_structure_tensor(oracle, block):
    """cv::cornerMinEigenVal (the response of goodFeaturesToTrack): Sobel derivatives, box sums over block x block
    (REFLECT_101 both), smaller eigenvalue of the 2 x 2 tensor.  The integer tensor sums come from scipy (sobel +
    an all-ones correlation, mode="mirror"); only calcMinEigenVal's closed form is restated here, in float32 as OpenCV
    has it.  (The oracle keeps the sums unscaled: goodFeaturesToTrack thresholds relative to the maximum.)"""
    rng = np.random.default_rng(block)
    im = rng.integers(0, 256, (37, 45), dtype=np.uint8)
    f = im.astype(np.float64)
    gx = np.rint(ndi.sobel(f, axis=1, mode="mirror")).astype(np.int64)
    gy = np.rint(ndi.sobel(f, axis=0, mode="mirror")).astype(np.int64)
    ones = np.ones((block, block), np.int64)
    sxx = ndi.correlate(gx * gx, ones, mode="mirror")
    sxy = ndi.correlate(gx * gy, ones, mode="mirror")
    syy = ndi.correlate(gy * gy, ones, mode="mirror")
    a, b, c = sxx.astype(np.float32) * np.float32(0.5), sxy.astype(np.float32), syy.astype(np.float32) * np.float32(0.5)
    t = a - c
    want = (a + c) - np.sqrt(t * t + b * b)
    assert np.array_equal(oracle.min_eig_map(im, block), want.astype(np.float32))

"""Known-answer tests of the oracle's OpenCV-primitive restatements (hand-computed) and agreement
with the independent numpy restatement (tests/pyref.py)."""
import numpy as np
import pytest

import pyref


def test_rng_first_outputs_by_hand(oracle):
    # state0 = 123: next = 123 * 4164903690 + 0
    s1 = 123 * 4164903690
    s2 = (s1 & 0xFFFFFFFF) * 4164903690 + (s1 >> 32)
    raw = oracle.rng_raw(2, 123)
    assert int(raw[0]) == s1 & 0xFFFFFFFF
    assert int(raw[1]) == s2 & 0xFFFFFFFF
    # seed 0 is replaced by 0xffffffff (cv::RNG::RNG(uint64))
    assert int(oracle.rng_raw(1, 0)[0]) == (0xFFFFFFFF * 4164903690) & 0xFFFFFFFF


def test_rng_matches_independent_restatement(oracle):
    assert np.array_equal(oracle.rng_raw(4096, 123), pyref.rng_raw(4096, 123))
    for lo, hi in ((-1.0, 1.0), (-32.0, 32.0), (-0.5, 0.5), (0.0, 7.0)):
        a = oracle.rng_fill_uniform(5000, lo, hi, 123)
        assert np.array_equal(a, pyref.rng_fill_uniform(5000, lo, hi, 123))
        assert a.min() >= lo and a.max() < hi


def test_rng_power_of_two_amplitude_is_scaled_unit_noise(oracle):
    # AddNoise(amount) (patchmatch.cpp:146-147) == unit noise * amount (patchmatch_gpu.cu:301) exactly
    unit = oracle.rng_fill_uniform(10000, -1.0, 1.0, 123)
    for amount in (32.0, 16.0, 8.0, 2.0, 0.5, 0.25):
        assert np.array_equal(oracle.rng_fill_uniform(10000, -amount, amount, 123), unit * np.float32(amount))


def test_sobel_known_answers(oracle):
    # horizontal ramp I = 3x: dx = 8*3 = 24 in the interior, dy = 0; reflect-101 makes border dx = 0
    im = np.tile((3 * np.arange(10)).astype(np.uint8), (6, 1))
    g = oracle.gradient_magnitude(im)
    assert np.all(g[:, 1:-1] == 24.0)
    assert np.all(g[:, 0] == 0.0) and np.all(g[:, -1] == 0.0)
    # single bright pixel of 100 at (3,3): neighbours see |dx|,|dy| in {100, 200}
    im = np.zeros((7, 7), np.uint8)
    im[3, 3] = 100
    g = oracle.gradient_magnitude(im)
    assert g[3, 3] == 0.0
    assert g[3, 2] == 200.0 and g[3, 4] == 200.0 and g[2, 3] == 200.0 and g[4, 3] == 200.0
    assert g[2, 2] == np.float32(np.sqrt(np.float32(20000.0)))
    assert np.all(g[0] == 0) and np.all(g[:, 0] == 0)


@pytest.mark.parametrize("shape", [(5, 7), (16, 16), (33, 20), (2, 9)])
def test_sobel_matches_independent_restatement(oracle, shape):
    rng = np.random.default_rng(7)
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(oracle.gradient_magnitude(im), pyref.gradient_magnitude(im))


def test_dilate(oracle):
    src = np.zeros((9, 11), np.float32)
    src[4, 5] = 3.0
    src[0, 0] = 7.0
    out = oracle.dilate_rect(src, 2)
    exp = np.zeros_like(src)
    exp[2:7, 3:8] = 3.0
    exp[0:3, 0:3] = 7.0
    assert np.array_equal(out, exp)
    rng = np.random.default_rng(3)
    src = rng.random((13, 17)).astype(np.float32)
    for k in (1, 3, 17):
        assert np.array_equal(oracle.dilate_rect(src, k), pyref.dilate_rect(src, k))


def test_rect_subpix_known_answers(oracle):
    src = np.arange(8 * 10, dtype=np.uint8).reshape(8, 10) * 2
    # integer centre + odd window = exact copy
    assert np.array_equal(oracle.get_rect_subpix(src, 3, 3, 4.0, 3.0), src[2:5, 3:6])
    # half-pixel shift: (s0 + s1) / 2 with round-half-up of the fixed-point sum: values 2n, 2n+2 -> 2n+1
    p = oracle.get_rect_subpix(src, 3, 1, 4.5, 3.0)
    assert np.array_equal(p[0], src[3, 3:6] + 1)
    # a = 0.25: (3*s0 + s1)/4 = s0 + 0.5 -> rounds up
    p = oracle.get_rect_subpix(src, 1, 1, 4.25, 3.0)
    assert p[0, 0] == src[3, 4] + 1
    # replicate border on the left / top
    p = oracle.get_rect_subpix(src, 3, 3, 0.0, 0.0)
    assert np.array_equal(p, src[np.ix_([0, 0, 1], [0, 0, 1])])
    # float path: exact lerp
    srcf = src.astype(np.float32)
    p = oracle.get_rect_subpix(srcf, 3, 1, 4.25, 3.0)
    assert np.array_equal(p[0], srcf[3, 3:6] * np.float32(0.75) + srcf[3, 4:7] * np.float32(0.25))


def test_rect_subpix_matches_independent_restatement(oracle):
    rng = np.random.default_rng(11)
    src8 = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    srcf = (rng.random((20, 30)) * 600).astype(np.float32)
    for _ in range(300):
        pw, ph = int(rng.choice([3, 5, 7, 11])), int(rng.choice([3, 5, 7, 11]))
        y = int(rng.integers(ph // 2, 20 - ph // 2))          # integer centre row, as on the path
        x = float(np.float32(rng.uniform(pw // 2, 30 - pw // 2 - 1)))
        for src in (src8, srcf):
            assert np.array_equal(oracle.get_rect_subpix(src, pw, ph, x, float(y)), pyref.rect_subpix(src, pw, ph, x, y))
    # last interior column / row with zero fraction goes through OpenCV's border branch
    for src in (src8, srcf):
        assert np.array_equal(oracle.get_rect_subpix(src, 5, 5, 27.0, 17.0), src[15:20, 25:30])


def test_gpu_get_subpixel(oracle):
    rng = np.random.default_rng(5)
    im = (rng.random((6, 9)) * 255).astype(np.float32)
    assert oracle.gpu_get_subpixel(im, 2.0, 3.0) == im[2, 3]
    assert oracle.gpu_get_subpixel(im, 2.0, 3.5) == np.float32(np.float32(0.5) * im[2, 3] + np.float32(0.5) * im[2, 4])
    for _ in range(200):
        r, c = float(np.float32(rng.uniform(0, 5))), float(np.float32(rng.uniform(0, 8)))
        assert oracle.gpu_get_subpixel(im, r, c) == pyref.gpu_get_subpixel(im, r, c)


def test_mean_from_sum_is_exact():
    """The device computes cv::mean's (float)(sum * (1./N)) without f64: p = s*hi, e = fma(s, hi, -p) + s*lo,
    mean = p + e (pm_device.hpp::mean_from_sum).  Exhaustive over every supported window and every possible sum."""
    for pw in range(3, 16, 2):
        for ph in range(3, 16, 2):
            n = pw * ph
            c = np.float64(1.0) / np.float64(n)
            hi = np.float32(c)
            lo = np.float32(c - np.float64(hi))
            s = np.arange(0, 255 * n + 1, dtype=np.int64)
            want = (s.astype(np.float64) * c).astype(np.float32)
            sf = s.astype(np.float32)
            p = (sf * hi).astype(np.float32)
            e1 = (sf.astype(np.float64) * np.float64(hi) - p.astype(np.float64)).astype(np.float32)  # the FMA, exact
            e2 = (sf * lo).astype(np.float32)
            got = (p + (e1 + e2).astype(np.float32)).astype(np.float32)
            assert np.array_equal(got, want), (pw, ph)


def test_lerp_weights_fit_16_bits_when_clamped():
    """The sweep kernels pack the fixed-point lerp weights a11 = rint((1 - a) * 65536), a12 = rint(a * 65536) into
    16 bits each (v_dot2_u32_u16) by clamping them to 65535 (pm_device.hpp::cpu_color_weights).  A weight of 65536
    only ever meets a weight of 0 or 1 (a within one ulp-of-1 of 2^-17), and for every such pair the clamped sum
    r0 * min(a11, 65535) + r1 * min(a12, 65535) + 2^15 has the same byte 2 -- the sample -- as the unclamped one, for
    all bytes r0, r1.  Pairs collected from every a = i / 2^24, the floats next to 0, 1 and the ties, random floats."""
    def weights(a):
        a = a.astype(np.float32)
        ia = (np.float32(1.0) - a).astype(np.float32)
        a11 = np.rint((ia * np.float32(65536.0)).astype(np.float32)).astype(np.int64)
        a12 = np.rint((a * np.float32(65536.0)).astype(np.float32)).astype(np.int64)
        return a11, a12

    chunks = [np.arange(0, 1 << 24, dtype=np.float64) / float(1 << 24)]
    near = [np.float32(0.0)]
    x = np.float32(0.0)
    for _ in range(64):
        x = np.nextafter(x, np.float32(1.0))
        near.append(x)
    x = np.float32(1.0)
    for _ in range(64):
        x = np.nextafter(x, np.float32(0.0))
        near.append(x)
    for e in (2.0 ** -17, 2.0 ** -16, 1 - 2.0 ** -17, 1 - 2.0 ** -16):
        c = np.float32(e)
        lo = hi = c
        near.append(c)
        for _ in range(64):
            lo, hi = np.nextafter(lo, np.float32(0.0)), np.nextafter(hi, np.float32(1.0))
            near += [lo, hi]
    chunks.append(np.array(near, np.float64))
    chunks.append(np.random.default_rng(3).random(1 << 20))
    extreme = set()
    for ch in chunks:
        a = ch[ch < 1.0]
        a11, a12 = weights(a)
        assert a11.min() >= 0 and a12.min() >= 0 and a11.max() <= 65536 and a12.max() <= 65536
        m = (a11 == 65536) | (a12 == 65536)
        extreme |= set(zip(a11[m].tolist(), a12[m].tolist()))
    assert (65536, 0) in extreme and (0, 65536) in extreme and len(extreme) <= 6, extreme
    r0, r1 = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    for a11, a12 in extreme:
        assert min(a11, a12) <= 1, (a11, a12)
        full = (r0 * a11 + r1 * a12 + (1 << 15)) >> 16
        clamped = (r0 * min(a11, 65535) + r1 * min(a12, 65535) + (1 << 15)) >> 16
        assert full.max() <= 255 and np.array_equal(full & 0xff, clamped & 0xff), (a11, a12)


def test_resize_linear_u8_known_answers(oracle):
    """cv::resize INTER_LINEAR on 8-bit images as both reference PatchMatch tests use it (size / 2,
    patchmatch_test.cpp:131-133): an exact 2x2 shrink is the rounded block mean; the general path is checked by hand
    on small cases (11-bit fixed-point weights; parity with OpenCV itself is unpinned)."""
    a = np.array([[10, 20, 30, 41], [12, 22, 33, 44], [0, 255, 255, 255], [1, 2, 254, 255]], np.uint8)
    half = oracle.resize_linear_u8(a, 2, 2)
    assert half.tolist() == [[(10 + 20 + 12 + 22 + 2) >> 2, (30 + 41 + 33 + 44 + 2) >> 2],
                             [(0 + 255 + 1 + 2 + 2) >> 2, (255 + 255 + 254 + 255 + 2) >> 2]]
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (480, 752), dtype=np.uint8)
    h = oracle.resize_linear_u8(img, 240, 376)
    want = (img[0::2, 0::2].astype(np.int32) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(h, want.astype(np.uint8))
    # identity size: every pixel keeps its value (weight 2048 / 0)
    assert np.array_equal(oracle.resize_linear_u8(img[:9, :13], 9, 13), img[:9, :13])
    # a constant image stays constant through the general path; a horizontal ramp 4 -> 3 columns by hand:
    # fx = (dx + 0.5) * 4/3 - 0.5 = 1/6, 1.5, 17/6 -> sx 0, 1, 2 with weights (1707, 341), (1024, 1024), (341, 1707)
    assert (oracle.resize_linear_u8(np.full((7, 10), 77, np.uint8), 5, 6) == 77).all()
    ramp = np.array([[0, 60, 120, 180]] * 2, np.uint8)
    got = oracle.resize_linear_u8(ramp, 2, 3)
    f = [np.float32((dx + 0.5) * (4.0 / 3.0) - 0.5) for dx in range(3)]
    w = [(int(np.rint((np.float32(1) - (v - np.floor(v))) * np.float32(2048))),
          int(np.rint((v - np.floor(v)) * np.float32(2048)))) for v in f]
    assert w == [(1707, 341), (1024, 1024), (341, 1707)]
    # vertical scale 1: fy = 0, weights (2048, 0): out = (((2048 * (h >> 4)) >> 16) + 0 + 2) >> 2
    hs = [0 * w[0][0] + 60 * w[0][1], 60 * w[1][0] + 120 * w[1][1], 120 * w[2][0] + 180 * w[2][1]]
    want = [(((2048 * (h >> 4)) >> 16) + 2) >> 2 for h in hs]
    assert want == [10, 90, 170]
    assert got.tolist() == [want, want]

"""Row f-3 of SURVEY.md section 8: disparity -> range -> range-dependent correction (include/pm/imaging.h).

The oracle (oracle/pm_imaging_oracle.c) restates StereoCamera::DispToDepth, imaging::RemoveBackscatter,
imaging::CorrectAttenuation, ComputeIntensity and imaging::FindDarkFast; the reference holds no expected outputs
for them (parity unpinned).  Integer / comparison results are checked exactly; float images to RTOL relative
(the device's expf and glibc's differ by at most a couple of ulp, amplified by the nested exponentials)."""
import numpy as np
import pytest

import oracle_lib as O

RTOL, ATOL = 1e-5, 1e-6

# the initial guesses of EnhanceUnderwater (src/vehicle/imaging/enhance.cpp:44-49) and a beta_D of the same scale
B0 = (0.132, 0.115, 0.0559)
BETA_B0 = (0.358, 0.695, 1.11)
X0 = (0.30, 0.25, 0.40, -0.20, -0.15, -0.30, 0.10, 0.12, 0.08, -0.05, -0.04, -0.06)


def scene(rows, cols, seed):
    rng = np.random.default_rng(seed)
    disp = rng.uniform(2.0, 96.0, (rows, cols)).astype(np.float32)
    disp[rng.random((rows, cols)) < 0.2] = 0.0  # masked-out pixels: no range
    bgr = rng.uniform(0.0, 1.0, (rows, cols, 3)).astype(np.float32)
    return bgr, disp


# ---- oracle known answers (CPU) ------------------------------------------------------------------------
def test_oracle_disp_to_range_known_values():
    d = np.array([[0.0, -1.0, 1.0, 2.0, 50.0, 0.5]], np.float32)
    r = O.disp_to_range(d, 400.0, 0.25)
    assert r.tolist() == [[0.0, 0.0, 100.0, 50.0, 2.0, 200.0]]
    # double division, then one rounding to float
    d = np.array([[3.0, 7.0]], np.float32)
    assert r.dtype == np.float32 and np.array_equal(O.disp_to_range(d, 415.876509, 0.12),
                                                    np.float32(415.876509 * 0.12 / d.astype(np.float64)))


def test_oracle_backscatter_and_attenuation_closed_form():
    bgr = np.full((1, 3, 3), 0.5, np.float32)
    rng = np.array([[0.0, 1.0, 4.0]], np.float32)
    out = O.remove_backscatter(bgr, rng, B0, BETA_B0)
    z = np.array([20.0, 1.0, 4.0])  # no range -> 20 m of water column
    for c in range(3):
        want = np.maximum(0.5 - B0[c] * (1.0 - np.exp(-BETA_B0[c] * z)), 0.0)
        assert np.allclose(out[0, :, c], want, rtol=1e-6)
    # range 0 with B = 1 removes everything (clamped at 0)
    assert O.remove_backscatter(bgr, rng, (1, 1, 1), (5, 5, 5))[0, 0].tolist() == [0.0, 0.0, 0.0]
    out = O.correct_attenuation(bgr, rng, X0)
    z = np.array([4.0, 1.0, 4.0])  # no range -> the largest range of the map
    for c in range(3):
        beta = X0[c] * np.exp(X0[3 + c] * z) + X0[6 + c] * np.exp(X0[9 + c] * z)
        assert np.allclose(out[0, :, c], 0.5 * np.exp(beta * z), rtol=1e-6)


def test_oracle_intensity_and_find_dark():
    bgr = np.zeros((2, 2, 3), np.float32)
    bgr[0, 0] = (1, 0, 0)
    bgr[0, 1] = (0, 1, 0)
    bgr[1, 0] = (0, 0, 1)
    bgr[1, 1] = (1, 1, 1)
    g = O.compute_intensity(bgr)
    assert g[0, 0] == np.float32(0.114) and g[0, 1] == np.float32(0.587) and g[1, 0] == np.float32(0.299)
    assert abs(g[1, 1] - 1.0) < 1e-6
    # uniform intensities: the 1 % threshold over pixels with range ends near 0.01
    rng = np.random.default_rng(3)
    inten = rng.uniform(0, 1, (200, 300)).astype(np.float32)
    rmap = np.ones((200, 300), np.float32)
    thr, mask = O.find_dark(inten, rmap, 0.01)
    assert 0.008 < thr < 0.012
    assert abs(int((mask > 0).sum()) - 600) < 60
    # pixels without range never count
    rmap[:, :150] = 0.05
    thr2, mask2 = O.find_dark(inten, rmap, 0.01)
    assert thr2 > thr and not mask2[:, :150].any()


# ---- device parity ----------------------------------------------------------------------------------------
def _dev(t, a):
    return t.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(48, 64), (37, 53), (1, 7), (240, 322)])
def test_device_stages_match_oracle(pm, rows, cols):
    import torch
    bgr, disp = scene(rows, cols, rows * 7 + cols)
    fx, baseline = 415.876509, 0.12
    with pm.Engine(pm.default_params(0, patch=5), max_rows=max(rows, 16), max_cols=max(cols, 16)) as e:
        d_disp, d_bgr = _dev(torch, disp), _dev(torch, bgr)
        d_range = torch.empty_like(d_disp)
        e.disp_to_range(d_disp.data_ptr(), rows, cols, fx, baseline, d_range.data_ptr())
        e.synchronize()
        want_range = O.disp_to_range(disp, fx, baseline)
        assert np.array_equal(d_range.cpu().numpy(), want_range), "DispToDepth is exact (double divide, one rounding)"

        d_out = torch.empty_like(d_bgr)
        e.remove_backscatter(d_bgr.data_ptr(), d_range.data_ptr(), rows, cols, B0, BETA_B0, d_out.data_ptr())
        e.synchronize()
        want_d = O.remove_backscatter(bgr, want_range, B0, BETA_B0)
        np.testing.assert_allclose(d_out.cpu().numpy(), want_d, rtol=RTOL, atol=ATOL)

        d_j = torch.empty_like(d_bgr)
        e.correct_attenuation(d_out.data_ptr(), d_range.data_ptr(), rows, cols, X0, d_j.data_ptr())
        e.synchronize()
        want_j = O.correct_attenuation(want_d, want_range, X0)
        np.testing.assert_allclose(d_j.cpu().numpy(), want_j, rtol=RTOL, atol=ATOL)

        # the fused pass equals the chain
        d_f, d_r2 = torch.empty_like(d_bgr), torch.empty_like(d_disp)
        e.range_enhance(d_bgr.data_ptr(), d_disp.data_ptr(), rows, cols, fx, baseline, B0, BETA_B0, X0,
                        d_r2.data_ptr(), d_f.data_ptr())
        e.synchronize()
        assert torch.equal(d_r2, d_range)
        assert torch.equal(d_f, d_j), "fused kernel = the three stages, bit for bit on the device"

        d_g = torch.empty_like(d_disp)
        e.compute_intensity(d_bgr.data_ptr(), rows, cols, d_g.data_ptr())
        e.synchronize()
        assert np.array_equal(d_g.cpu().numpy(), O.compute_intensity(bgr)), "BGR2GRAY: three products, two sums"

        d_mask = torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
        thr = e.find_dark(d_g.data_ptr(), d_range.data_ptr(), rows, cols, 0.05, d_mask.data_ptr())
        want_thr, want_mask = O.find_dark(O.compute_intensity(bgr), want_range, 0.05)
        assert thr == want_thr and np.array_equal(d_mask.cpu().numpy(), want_mask)


@pytest.mark.gpu
def test_unaligned_pointers_and_errors(pm):
    import torch
    rows, cols = 20, 33
    bgr, disp = scene(rows, cols, 5)
    with pm.Engine(pm.default_params(0, patch=5), max_rows=32, max_cols=64) as e:
        # views that start 4 bytes into an allocation: the kernels must not assume 16-byte alignment
        buf_b = torch.zeros(rows * cols * 3 + 1, device="cuda")
        buf_r = torch.zeros(rows * cols + 1, device="cuda")
        buf_o = torch.zeros(rows * cols * 3 + 1, device="cuda")
        buf_b[1:] = _dev(torch, bgr).reshape(-1)
        rng = O.disp_to_range(disp, 400.0, 0.1)
        buf_r[1:] = _dev(torch, rng).reshape(-1)
        e.remove_backscatter(buf_b[1:].data_ptr(), buf_r[1:].data_ptr(), rows, cols, B0, BETA_B0, buf_o[1:].data_ptr())
        e.synchronize()
        got = buf_o[1:].cpu().numpy().reshape(rows, cols, 3)
        np.testing.assert_allclose(got, O.remove_backscatter(bgr, rng, B0, BETA_B0), rtol=RTOL, atol=ATOL)
        with pytest.raises(pm.PmError) as err:
            e.disp_to_range(0, rows, cols, 1.0, 1.0, buf_r.data_ptr())
        assert err.value.status == pm.PM_ERR_INVALID_ARG


@pytest.mark.gpu
def test_stereo_to_corrected_image_without_host_round_trip(pm, oracle, synth):
    """Match() -> disparity -> range -> correction, all enqueued on the handle's stream."""
    import torch
    rows, cols = 96, 160
    p = synth.make_pair(3, rows, cols)
    rng = np.random.default_rng(1)
    bgr = rng.uniform(0, 1, (rows, cols, 3)).astype(np.float32)
    with pm.Engine(pm.default_params(0, patch=5, patchmatch_iters=2), max_rows=rows, max_cols=cols) as e:
        L, R = _dev(torch, p["left"]), _dev(torch, p["right"])
        SL, SR = _dev(torch, p["seed_l"]), _dev(torch, p["seed_r"])
        DL, DR = torch.empty_like(SL), torch.empty_like(SR)
        d_bgr, d_out = _dev(torch, bgr), torch.empty((rows, cols, 3), device="cuda")
        e.match_device(1, L.data_ptr(), R.data_ptr(), rows, cols, SL.data_ptr(), SR.data_ptr(), DL.data_ptr(),
                       DR.data_ptr())
        e.range_enhance(d_bgr.data_ptr(), DL.data_ptr(), rows, cols, 400.0, 0.1, B0, BETA_B0, X0, None,
                        d_out.data_ptr())
        e.synchronize()
    el, _ = oracle.match(oracle.default_params(0, patch=5, n_iters=2, nthreads=8), p["left"], p["right"],
                         p["seed_l"], p["seed_r"])
    r = O.disp_to_range(el, 400.0, 0.1)
    want = O.correct_attenuation(O.remove_backscatter(bgr, r, B0, BETA_B0), r, X0)
    np.testing.assert_allclose(d_out.cpu().numpy(), want, rtol=RTOL, atol=ATOL)

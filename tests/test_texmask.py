"""ForegroundTextureMask (src/vehicle/stereo_matching/patchmatch.cpp:19-49): known answers of the oracle's restatement and,
with -m gpu, bit parity of the device version (pm_foreground_texture_mask) with it.  Nothing in the reference calls the
function; it completes stereo_matching/patchmatch.{hpp,cpp} (missing 4)."""
import numpy as np
import pytest


def test_known_answers(oracle):
    flat = np.full((24, 40), 77, np.uint8)
    assert not oracle.foreground_texture_mask(flat, 4, 0.0, 1).any()       # no texture: gradient 0, never > 0
    # one bright pixel: the gradient is 200 wherever the (2k+1)^2 rectangle reaches it, 0 elsewhere
    im = np.zeros((24, 40), np.uint8)
    im[10, 20] = 200
    m = oracle.foreground_texture_mask(im, 3, 100.0, 1)
    want = np.zeros_like(im)
    want[7:14, 17:24] = 255
    assert np.array_equal(m, want)
    assert not oracle.foreground_texture_mask(im, 3, 200.0, 1).any()       # strictly greater than min_grad
    # a step edge: a band of 2k + 1 columns... minus one (the rectangle must contain both sides)
    im = np.zeros((16, 32), np.uint8)
    im[:, 16:] = 90
    m = oracle.foreground_texture_mask(im, 2, 50.0, 1)
    assert np.array_equal(np.flatnonzero(m[5]), np.arange(14, 18))
    # the window is clipped at the border: a corner pixel still sees its neighbours
    im = np.zeros((12, 12), np.uint8)
    im[0, 0] = 255
    m = oracle.foreground_texture_mask(im, 2, 1.0, 1)
    assert m[:3, :3].all() and not m[3:, :].any() and not m[:, 3:].any()
    # the arguments the reference CHECK-fails on (patchmatch.cpp:25-27)
    for ksize, down in ((4, 0), (4, 9), (2, 2), (1, 1)):
        with pytest.raises(ValueError):
            oracle.foreground_texture_mask(im, ksize, 1.0, down)


def test_downsized_path_against_its_parts(oracle, synth):
    """downsize > 1 = resize -> gradient on the small image -> threshold -> resize back: composed here from the
    oracle's own resize and the downsize == 1 path."""
    p = synth.make_pair(3, rows=96, cols=160)
    g = p["left"].copy()
    g[:, :70] = 120  # a flat region: no texture there
    for down, ksize in ((2, 6), (3, 9), (4, 12)):
        small = oracle.resize_linear_u8(g, g.shape[0] // down, g.shape[1] // down)
        m_small = oracle.foreground_texture_mask(small, ksize // down, 20.0, 1)
        want = oracle.resize_linear_u8(m_small, g.shape[0], g.shape[1])
        got = oracle.foreground_texture_mask(g, ksize, 20.0, down)
        assert np.array_equal(got, want)
        assert 0 < (got > 0).mean() < 1


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,ksize,min_grad,down", [(97, 161, 4, 12.0, 1), (96, 160, 6, 25.0, 2), (120, 200, 9, 8.5, 3),
                                                            (720, 1280, 12, 20.0, 4), (64, 96, 16, 30.0, 8), (33, 47, 2, 0.0, 1)])
def test_device_mask_equals_the_oracle(pm, oracle, synth, rows, cols, ksize, min_grad, down):
    torch = pytest.importorskip("torch")
    p = synth.make_pair(rows, rows=rows, cols=cols)
    g = np.ascontiguousarray(p["left"])
    dev = torch.device("cuda:0")
    dg = torch.from_numpy(g).to(dev)
    dm = torch.zeros((rows, cols), dtype=torch.uint8, device=dev)
    with pm.Engine(pm.default_params(0), max_rows=rows, max_cols=cols) as e:
        e.foreground_texture_mask(dg.data_ptr(), rows, cols, ksize, min_grad, down, dm.data_ptr())
        e.synchronize()
        with pytest.raises(pm.PmError) as err:   # what the reference CHECK-fails on
            e.foreground_texture_mask(dg.data_ptr(), rows, cols, 2, 1.0, 2, dm.data_ptr())
        assert err.value.status == pm.PM_ERR_INVALID_ARG
    assert np.array_equal(dm.cpu().numpy(), oracle.foreground_texture_mask(g, ksize, min_grad, down))

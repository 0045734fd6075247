"""Seeded synthetic rectified stereo pairs (SURVEY.md 8d "Synthetic inputs").

Pair i uses numpy's PCG64 seeded with 1000+i.  Everything is generated in left-image
coordinates:
  * right image  = 4-octave value-noise texture + per-pixel grain, u8 in [16, 240];
  * ground truth = smooth slanted background (d in [2, 12]) plus 3-6 slanted rectangular / elliptic
                   foreground patches (d in [16, 96]; all scaled by d_max/96 for small images), i.e. inside the reference seeder's reachable
                   range [1, 98] for max_disp 128 / templ_cols 31 (stereo_matcher.cpp:69-92);
  * left image   = bilinear warp Il(x, y) = Ir(x - d(x, y), y), + N(0, 1) sensor noise on both;
  * seed maps    = what SparseInit would hand the engine (patchmatch_gpu.cu:414-442): ground truth
                   rounded to integer pixels at ~200 jittered grid points, dilated with the
                   (2*(2^f+1)+1)^2 rectangle, one map per view.
No dataset, no file I/O: the reference's fixtures never travel to the GPU box.
"""
import numpy as np
from scipy.ndimage import maximum_filter


def _value_noise(rng, rows, cols, cell):
    gh, gw = rows // cell + 2, cols // cell + 2
    grid = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(rows, dtype=np.float32) / cell
    xs = np.arange(cols, dtype=np.float32) / cell
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    ty = (ys - y0)[:, None]
    tx = (xs - x0)[None, :]
    ty = ty * ty * (3 - 2 * ty)
    tx = tx * tx * (3 - 2 * tx)
    g00 = grid[y0][:, x0]
    g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]
    g11 = grid[y0 + 1][:, x0 + 1]
    return (g00 * (1 - tx) + g01 * tx) * (1 - ty) + (g10 * (1 - tx) + g11 * tx) * ty


def texture(rng, rows, cols):
    t = np.zeros((rows, cols), np.float32)
    amp = 1.0
    for cell in (32, 16, 8, 4):
        t += amp * _value_noise(rng, rows, cols, cell)
        amp *= 0.6
    t += 0.35 * rng.random((rows, cols)).astype(np.float32)  # pixel-level grain
    t -= t.min()
    t /= max(float(t.max()), 1e-6)
    return 16.0 + 224.0 * t


def ground_truth(rng, rows, cols, d_max=96.0):
    ys, xs = np.mgrid[0:rows, 0:cols].astype(np.float32)
    u, v = xs / cols, ys / rows
    k = d_max / 96.0
    d = k * (2.0 + 6.0 * u * rng.random() + 4.0 * v * rng.random())
    for _ in range(int(rng.integers(3, 7))):
        cx, cy = rng.uniform(0.15, 0.85) * cols, rng.uniform(0.15, 0.85) * rows
        hw, hh = rng.uniform(0.06, 0.22) * cols, rng.uniform(0.08, 0.28) * rows
        base = rng.uniform(16.0 * k, 84.0 * k)
        sx, sy = rng.uniform(-10.0, 10.0) * k, rng.uniform(-6.0, 6.0) * k
        plane = base + sx * (xs - cx) / max(hw, 1.0) * 0.5 + sy * (ys - cy) / max(hh, 1.0) * 0.5
        if rng.random() < 0.5:
            inside = (np.abs(xs - cx) <= hw) & (np.abs(ys - cy) <= hh)
        else:
            inside = ((xs - cx) / hw) ** 2 + ((ys - cy) / hh) ** 2 <= 1.0
        d = np.where(inside & (plane > d), plane, d)
    return np.clip(d, 1.0, d_max).astype(np.float32)


def _warp_left(right_f, disp):
    rows, cols = disp.shape
    xs = np.arange(cols, dtype=np.float32)[None, :] - disp
    xs = np.clip(xs, 0.0, cols - 1.0)
    x0 = np.floor(xs).astype(np.int64)
    x1 = np.minimum(x0 + 1, cols - 1)
    t = xs - x0
    rr = np.arange(rows)[:, None]
    return right_f[rr, x0] * (1 - t) + right_f[rr, x1] * t


def seed_maps(rng, disp, n_points=200, dilate_factor=4):
    """(seed_l, seed_r): sparse integer disparities dilated as SparseInit does."""
    rows, cols = disp.shape
    k = int(2 ** dilate_factor) + 1  # dilate_size, patchmatch_gpu.cu:436
    gy = max(1, int(round(np.sqrt(n_points * rows / cols))))
    gx = max(1, int(round(n_points / gy)))
    sl = np.zeros((rows, cols), np.float32)
    sr = np.zeros((rows, cols), np.float32)
    for iy in range(gy):
        for ix in range(gx):
            y = int((iy + rng.uniform(0.2, 0.8)) * rows / gy)
            x = int((ix + rng.uniform(0.2, 0.8)) * cols / gx)
            d = float(np.rint(disp[y, x]))
            if d < 1.0 or x - d < 0:
                continue
            sl[y, x] = d
            sr[y, int(x - d)] = max(sr[y, int(x - d)], d)
    size = 2 * k + 1
    sl = maximum_filter(sl, size=size, mode="constant", cval=0.0)
    sr = maximum_filter(sr, size=size, mode="constant", cval=0.0)
    return sl.astype(np.float32), sr.astype(np.float32)


def make_pair(index, rows=720, cols=1280, d_max=96.0, n_points=200, dilate_factor=4):
    """Returns dict(left, right (u8), gt (f32, left view), seed_l, seed_r (f32))."""
    rng = np.random.default_rng(1000 + int(index))
    d_max = float(min(d_max, max(4.0, cols / 4.0)))
    right_f = texture(rng, rows, cols)
    gt = ground_truth(rng, rows, cols, d_max)
    left_f = _warp_left(right_f, gt)
    left = np.clip(np.rint(left_f + rng.normal(0.0, 1.0, left_f.shape)), 0, 255).astype(np.uint8)
    right = np.clip(np.rint(right_f + rng.normal(0.0, 1.0, right_f.shape)), 0, 255).astype(np.uint8)
    seed_l, seed_r = seed_maps(rng, gt, n_points, dilate_factor)
    return {"left": left, "right": right, "gt": gt, "seed_l": seed_l, "seed_r": seed_r}


def to_bgr(gray, seed):
    """A synthetic underwater-looking BGR image whose stereo-ready enhancement is well defined: the gray pattern
    under a smooth coloured illuminant (blue-green cast, vignette).  BASELINE configs[4]'s input shape."""
    rows, cols = gray.shape
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    vig = 0.55 + 0.45 * np.exp(-(((xx - cols / 2) / (0.6 * cols)) ** 2 + ((yy - rows / 2) / (0.6 * rows)) ** 2))
    g = gray.astype(np.float32)
    rng = np.random.default_rng(seed)
    gains = (1.0, 0.85, 0.55)  # B, G, R
    bgr = np.stack([np.clip(g * gains[c] * vig + rng.normal(0, 0.6, g.shape), 0, 255) for c in range(3)], -1)
    return np.rint(bgr).astype(np.uint8)

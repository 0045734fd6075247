"""Independent numpy / pure-Python restatement of the reference path, for SMALL cases only.

Written separately from oracle/pm_oracle.c (different structure: vectorised patches, clamped
indexing instead of OpenCV's adjustRect code path) so that agreement between the two is evidence
that the C oracle says what its comments claim.  All float arithmetic is done in np.float32, one
rounding per operation.  Citations are relative to /root/reference.
"""
import numpy as np

f32 = np.float32


# ---- OpenCV primitives -------------------------------------------------------------------------
def rng_raw(n, seed=123):
    """cv::RNG multiply-with-carry: state = lo32(state) * 4164903690 + hi32(state)."""
    state = seed if seed else 0xFFFFFFFF
    out = []
    for _ in range(n):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(state & 0xFFFFFFFF)
    return np.array(out, np.uint32)


def rng_fill_uniform(n, lo, hi, seed=123):
    raw = rng_raw(n, seed).astype(np.int64)
    signed = np.where(raw >= 2 ** 31, raw - 2 ** 32, raw).astype(np.float32)
    scale = f32((hi - lo) * 2.0 ** -32)
    shift = f32((hi + lo) * 0.5)
    return (signed * scale).astype(np.float32) + shift


def gradient_magnitude(im):
    p = np.pad(im.astype(np.int64), 1, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    dx = (p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])
    dy = (p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])
    s = (dx.astype(np.float32) ** 2 + dy.astype(np.float32) ** 2).astype(np.float32)
    return np.sqrt(s).astype(np.float32)


def dilate_rect(src, k):
    rows, cols = src.shape
    out = np.empty_like(src)
    for y in range(rows):
        for x in range(cols):
            out[y, x] = src[max(0, y - k):y + k + 1, max(0, x - k):x + k + 1].max()
    return out


def cv_round(v):
    return np.rint(np.asarray(v, np.float32)).astype(np.int64)  # half to even


def sat_u8(v):
    return np.clip(cv_round(v), 0, 255)


def rect_subpix(src, pw, ph, cx, cy):
    """getRectSubPix through clamped (replicated) indices.  Equals OpenCV's result whenever the
    window is inside the image or touches the border with a zero fractional part there, which is
    every case the PatchMatch path produces."""
    rows, cols = src.shape
    cx = f32(f32(cx) - f32(pw - 1) * f32(0.5))
    cy = f32(f32(cy) - f32(ph - 1) * f32(0.5))
    ipx, ipy = int(np.floor(cx)), int(np.floor(cy))
    a, b = f32(cx - f32(ipx)), f32(cy - f32(ipy))
    ia, ib = f32(f32(1) - a), f32(f32(1) - b)
    ys0 = np.clip(ipy + np.arange(ph), 0, rows - 1)
    ys1 = np.clip(ipy + np.arange(ph) + 1, 0, rows - 1)
    xs0 = np.clip(ipx + np.arange(pw), 0, cols - 1)
    xs1 = np.clip(ipx + np.arange(pw) + 1, 0, cols - 1)
    s00, s01 = src[np.ix_(ys0, xs0)], src[np.ix_(ys0, xs1)]
    s10, s11 = src[np.ix_(ys1, xs0)], src[np.ix_(ys1, xs1)]
    if src.dtype == np.uint8:
        w = [int(cv_round(f32(f32(x) * f32(65536)))) for x in (ia * ib, a * ib, ia * b, a * b)]
        acc = (s00.astype(np.int64) * w[0] + s01.astype(np.int64) * w[1] + s10.astype(np.int64) * w[2] +
               s11.astype(np.int64) * w[3])
        return ((acc + (1 << 15)) >> 16).astype(np.uint8)
    w = [f32(ia * ib), f32(a * ib), f32(ia * b), f32(a * b)]
    acc = (s00 * w[0]).astype(np.float32)
    acc = (acc + (s01 * w[1]).astype(np.float32)).astype(np.float32)
    acc = (acc + (s10 * w[2]).astype(np.float32)).astype(np.float32)
    acc = (acc + (s11 * w[3]).astype(np.float32)).astype(np.float32)
    return acc


# ---- SEM_CPU -----------------------------------------------------------------------------------
def cpu_functor(pl, pr, gl, gr, alpha=0.7, tau_color=50.0, tau_grad=20.0):
    """L1GradientCostFunction (test/stereo_matching/patchmatch_test.cpp:30-45)."""
    n = pl.size
    sc = int(np.abs(pl.astype(np.int64) - pr.astype(np.int64)).sum())
    sg = int(np.abs(sat_u8(gl) - sat_u8(gr)).sum())
    mc = f32(np.float64(sc) * (np.float64(1.0) / np.float64(n)))
    mg = f32(np.float64(sg) * (np.float64(1.0) / np.float64(n)))
    ec = min(mc, f32(tau_color))
    eg = min(mg, f32(tau_grad))
    al = f32(alpha)
    return f32(f32(al * ec) + f32(f32(f32(1) - al) * eg))


def cpu_cost(il, ir, gl, gr, pw, ph, x, y, d):
    xr = f32(f32(x) - f32(d))
    return cpu_functor(rect_subpix(il, pw, ph, x, y), rect_subpix(ir, pw, ph, xr, y),
                       rect_subpix(gl, pw, ph, x, y), rect_subpix(gr, pw, ph, xr, y))


def cpu_add_noise(disp, amount, mask=None, seed=123):
    noise = rng_fill_uniform(disp.size, -float(amount), float(amount), seed).reshape(disp.shape)
    out = disp.astype(np.float32).copy()
    sel = np.ones(disp.shape, bool) if mask is None else (mask != 0)
    out[sel] = (out[sel] + noise[sel]).astype(np.float32)
    return np.maximum(out, f32(0))


def _pn(il, ir, gl, gr, disp, x, y, pw, ph, xo, yo):
    """PropagateNeighbors (src/vehicle/stereo_matching/patchmatch.cpp:158-196)."""
    d0 = f32(min(max(disp[y, x], f32(0)), f32(f32(x) - f32(pw // 2))))
    dl = disp[y + yo, x + xo]
    best, c0 = d0, cpu_cost(il, ir, gl, gr, pw, ph, x, y, d0)
    if f32(f32(x) - dl) >= f32(pw // 2):
        if cpu_cost(il, ir, gl, gr, pw, ph, x, y, dl) < c0:
            best = dl
    disp[y, x] = best


def _skip(x, y, w, h, pw, ph):
    return y < ph // 2 or x < pw // 2 or y > h - ph // 2 - 1 or x > w - pw // 2 - 1


def cpu_propagate(il, ir, gl, gr, disp, ph, pw, pass_mask=15):
    """Patchmatch::Propagate (patchmatch.cpp:248-311), raster order exactly as written."""
    disp = disp.astype(np.float32).copy()
    h, w = il.shape
    if pass_mask & 1:
        for y in range(1, h):
            for x in range(1, w):
                if not _skip(x, y, w, h, pw, ph):
                    _pn(il, ir, gl, gr, disp, x, y, pw, ph, -1, 0)
    if pass_mask & 2:
        for y in range(1, h):
            for x in range(1, w):
                if not _skip(x, y, w, h, pw, ph):
                    _pn(il, ir, gl, gr, disp, x, y, pw, ph, 0, -1)
    if pass_mask & 4:
        for y in range(h - 2, -1, -1):
            for x in range(w - 2, -1, -1):
                if not _skip(x, y, w, h, pw, ph):
                    _pn(il, ir, gl, gr, disp, x, y, pw, ph, 1, 0)
    if pass_mask & 8:
        for y in range(h - 2, -1, -1):
            for x in range(w - 2, -1, -1):
                if not _skip(x, y, w, h, pw, ph):
                    _pn(il, ir, gl, gr, disp, x, y, pw, ph, 0, 1)
    return disp


def cpu_remove_background(il, ir, gl, gr, disp, ph, pw, factor=1.5):
    """Patchmatch::RemoveBackground (patchmatch.cpp:314-360)."""
    disp = disp.astype(np.float32).copy()
    h, w = il.shape
    for y in range(1, h):
        for x in range(1, w):
            if _skip(x, y, w, h, pw, ph):
                continue
            d0 = f32(min(max(disp[y, x], f32(0)), f32(f32(x) - f32(pw // 2))))
            c = cpu_cost(il, ir, gl, gr, pw, ph, x, y, d0)
            cb = cpu_cost(il, ir, gl, gr, pw, ph, x, y, f32(0))
            if c > f32(cb / f32(factor)):
                disp[y, x] = 0
    return disp


# ---- SEM_GPU -----------------------------------------------------------------------------------
def gpu_get_subpixel(im, row, col):
    """GetSubpixel (src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:18-42)."""
    row, col = f32(row), f32(col)
    r0, r1 = int(np.floor(row)), int(np.ceil(row))
    c0, c1 = int(np.floor(col)), int(np.ceil(col))
    trow, tcol = f32(row - f32(r0)), f32(col - f32(c0))
    v = lambda r, c: f32(im[r, c])
    a0 = f32(f32(f32(f32(1) - trow) * v(r0, c0)) + f32(trow * v(r1, c0)))
    a1 = f32(f32(f32(f32(1) - trow) * v(r0, c1)) + f32(trow * v(r1, c1)))
    return f32(f32(f32(f32(1) - tcol) * a0) + f32(tcol * a1))


def gpu_cost5(il, ir, gl, gr, yl, xl, yr, xr, alpha=0.9):
    """L1GradientCost3x3 (patchmatch_gpu.cu:72-114); il/ir are the u8 images (converted exactly)."""
    al = f32(alpha)
    cost = f32(0)
    for dy, dx in ((-1, -1), (-1, 1), (0, 0), (1, -1), (1, 1)):
        e0 = abs(f32(f32(il[yl + dy, xl + dx]) - gpu_get_subpixel(ir, f32(yr + dy), f32(f32(xr) + f32(dx)))))
        e1 = abs(f32(gl[yl + dy, xl + dx] - gpu_get_subpixel(gr, f32(yr + dy), f32(f32(xr) + f32(dx)))))
        cost = f32(cost + f32(f32(al * e0) + f32(f32(f32(1) - al) * e1)))
    return cost


def gpu_add_foreground_noise(disp, unit, scale):
    d = disp.astype(np.float32)
    mask = (d > 0).astype(np.float32)
    out = ((unit * f32(scale)).astype(np.float32) + d).astype(np.float32)
    out = (out * mask).astype(np.float32)
    return np.maximum(out, f32(0)) + f32(0)  # +0 canonicalises -0


def gpu_propagate(il, ir, gl, gr, disp, axis, direction, alpha=0.9):
    """PropagateRow / PropagateCol (patchmatch_gpu.cu:116-230) as a single stripe."""
    disp = disp.astype(np.float32).copy()
    h, w = il.shape
    r = 1
    if axis == 0:
        lo, hi = r, w - r - 1
        seq = range(lo, hi) if direction > 0 else range(hi, lo, -1)
        for row in range(r, h - r):
            for col in seq:
                x = f32(col)
                d0, d1 = disp[row, col], disp[row, col - direction]
                c0 = gpu_cost5(il, ir, gl, gr, row, col, row, max(f32(x - d0), f32(r)), alpha)
                c1 = gpu_cost5(il, ir, gl, gr, row, col, row, max(f32(x - d1), f32(r)), alpha)
                if c1 < c0:
                    disp[row, col] = min(d1, f32(x - f32(r)))
    else:
        lo, hi = r, h - r - 1
        seq = range(lo, hi) if direction > 0 else range(hi, lo, -1)
        for col in range(r, w - r):
            x = f32(col)
            for row in seq:
                d0, d1 = disp[row, col], disp[row - direction, col]
                c0 = gpu_cost5(il, ir, gl, gr, row, col, row, max(f32(x - d0), f32(r)), alpha)
                c1 = gpu_cost5(il, ir, gl, gr, row, col, row, max(f32(x - d1), f32(r)), alpha)
                if c1 < c0:
                    disp[row, col] = min(d1, f32(x - f32(r)))
    return disp


def gpu_mask_background(il, ir, gl, gr, disp, alpha=0.9, improve=0.8):
    disp = disp.astype(np.float32).copy()
    h, w = il.shape
    for row in range(1, h - 1):
        for col in range(1, w - 1):
            x = f32(col)
            c0 = gpu_cost5(il, ir, gl, gr, row, col, row, x, alpha)
            c1 = gpu_cost5(il, ir, gl, gr, row, col, row, max(f32(x - disp[row, col]), f32(1)), alpha)
            if not (c1 < f32(f32(improve) * c0)):
                disp[row, col] = 0
    return disp


def gpu_mask_occlusions(displ, dispr):
    out = displ.astype(np.float32).copy()
    h, w = out.shape
    for y in range(h):
        for x in range(w):
            dl = out[y, x]
            dr = dispr[y, int(max(f32(f32(x) - dl), f32(0)))]
            if float(dr) > 1.4 * float(dl) or float(dr) < 0.7 * float(dl):
                out[y, x] = 0
    return out

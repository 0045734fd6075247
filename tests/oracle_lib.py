"""ctypes binding of oracle/libpm_oracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libpm_oracle.so")
PMO_MAX_ITERS = 16
SEM_CPU, SEM_GPU = 0, 1


class Images(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("il", C.c_void_p), ("ir", C.c_void_p),
                ("gl", C.c_void_p), ("gr", C.c_void_p)]


class Functor(C.Structure):
    _fields_ = [("alpha", C.c_float), ("tau_color", C.c_float), ("tau_grad", C.c_float)]


class SeedParams(C.Structure):
    _fields_ = [("max_features", C.c_int), ("min_distance", C.c_int), ("quality_level", C.c_double),
                ("block_size", C.c_int), ("templ_cols", C.c_int), ("templ_rows", C.c_int), ("max_disp", C.c_int),
                ("max_matching_cost", C.c_double), ("use_harris", C.c_int), ("harris_k", C.c_double),
                ("subpixel_corners", C.c_int), ("subpix_winsize", C.c_int), ("subpix_zerozone", C.c_int),
                ("subpix_maxiters", C.c_int), ("subpix_epsilon", C.c_float), ("subpixel_refinement", C.c_int)]


class Params(C.Structure):
    _fields_ = [
        ("semantics", C.c_int), ("n_iters", C.c_int),
        ("noise_amp", C.c_float * PMO_MAX_ITERS),
        ("patch_w", C.c_int * PMO_MAX_ITERS), ("patch_h", C.c_int * PMO_MAX_ITERS),
        ("bg_patch_w", C.c_int), ("bg_patch_h", C.c_int), ("bg_factor", C.c_float),
        ("cost_alpha", C.c_float), ("functor", Functor), ("noise_seed", C.c_uint64),
        ("left_right_check", C.c_int), ("literal", C.c_int), ("nthreads", C.c_int),
    ]


_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.pmo_params_default.argtypes = [C.POINTER(Params), C.c_int]
    lib.pmo_rng_fill_uniform.argtypes = [vp, C.c_size_t, C.c_double, C.c_double, C.c_uint64]
    lib.pmo_rng_raw.argtypes = [vp, C.c_size_t, C.c_uint64]
    lib.pmo_gradient_magnitude.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.pmo_dilate_rect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.pmo_get_rect_subpix_u8.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]
    lib.pmo_get_rect_subpix_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]
    lib.pmo_cpu_functor.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(Functor)]
    lib.pmo_cpu_functor.restype = C.c_float
    lib.pmo_cpu_cost_literal.argtypes = [C.POINTER(Images), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.POINTER(Functor)]
    lib.pmo_cpu_cost_literal.restype = C.c_float
    lib.pmo_cpu_cost_direct.argtypes = [C.POINTER(Images), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.POINTER(Functor)]
    lib.pmo_cpu_cost_direct.restype = C.c_float
    lib.pmo_cpu_add_noise.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp, C.c_uint64]
    lib.pmo_cpu_propagate.argtypes = [C.POINTER(Images), vp, C.c_int, C.c_int, C.POINTER(Functor), C.c_int, C.c_int,
                                      C.c_int]
    lib.pmo_cpu_remove_background.argtypes = [C.POINTER(Images), vp, C.c_int, C.c_int, C.POINTER(Functor), C.c_float,
                                              C.c_int, C.c_int]
    lib.pmo_gpu_get_subpixel.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float]
    lib.pmo_gpu_get_subpixel.restype = C.c_float
    lib.pmo_gpu_cost5.argtypes = [C.POINTER(Images), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    lib.pmo_gpu_cost5.restype = C.c_float
    lib.pmo_gpu_add_foreground_noise.argtypes = [vp, vp, C.c_size_t, C.c_float]
    lib.pmo_gpu_propagate_row.argtypes = [C.POINTER(Images), vp, C.c_int, C.c_int, C.c_float, C.c_int]
    lib.pmo_gpu_propagate_col.argtypes = [C.POINTER(Images), vp, C.c_int, C.c_int, C.c_float, C.c_int]
    lib.pmo_gpu_mask_background.argtypes = [C.POINTER(Images), vp, C.c_int, C.c_float, C.c_float, C.c_int]
    lib.pmo_gpu_mask_occlusions.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.pmo_match_view.argtypes = [C.POINTER(Params), C.POINTER(Images), vp]
    lib.pmo_match.argtypes = [C.POINTER(Params), vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.pmo_flip_h_u8.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.pmo_resize_linear_u8.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    lib.pmo_flip_h_f32.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.pmo_seed_params_default.argtypes = [C.POINTER(SeedParams)]
    lib.pmo_seed_params_default.restype = None
    lib.pmo_min_eig_map.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.pmo_min_eig_map.restype = None
    lib.pmo_gftt_detect.argtypes = [vp, C.c_int, C.c_int, C.POINTER(SeedParams), vp, vp, C.c_int]
    lib.pmo_gftt_detect.restype = C.c_int
    lib.pmo_match_rectified.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(SeedParams)]
    lib.pmo_match_rectified.restype = C.c_double
    lib.pmo_sparse_init.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(SeedParams), vp]
    lib.pmo_sparse_init.restype = None
    for name in ("pmo_params_default", "pmo_rng_fill_uniform", "pmo_rng_raw", "pmo_gradient_magnitude",
                 "pmo_dilate_rect", "pmo_get_rect_subpix_u8", "pmo_get_rect_subpix_f32", "pmo_cpu_add_noise",
                 "pmo_cpu_propagate", "pmo_cpu_remove_background", "pmo_gpu_add_foreground_noise",
                 "pmo_gpu_propagate_row", "pmo_gpu_propagate_col", "pmo_gpu_mask_background",
                 "pmo_gpu_mask_occlusions", "pmo_match_view", "pmo_match", "pmo_flip_h_u8", "pmo_flip_h_f32",
                 "pmo_resize_linear_u8"):
        getattr(lib, name).restype = None
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def c_u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def c_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def functor(alpha=0.7, tau_color=50.0, tau_grad=20.0):
    return Functor(alpha, tau_color, tau_grad)


def default_params(semantics=SEM_CPU, patch=None, **kw):
    p = Params()
    load().pmo_params_default(C.byref(p), semantics)
    if patch is not None:
        for i in range(PMO_MAX_ITERS):
            p.patch_w[i] = patch
            p.patch_h[i] = patch
        p.bg_patch_w = patch
        p.bg_patch_h = patch
    for k, v in kw.items():
        if k in ("noise_amp", "patch_w", "patch_h"):
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


def rng_fill_uniform(n, lo, hi, seed=123):
    out = np.empty(n, np.float32)
    load().pmo_rng_fill_uniform(_p(out), n, lo, hi, seed)
    return out


def rng_raw(n, seed=123):
    out = np.empty(n, np.uint32)
    load().pmo_rng_raw(_p(out), n, seed)
    return out


def gradient_magnitude(im):
    im = c_u8(im)
    g = np.empty(im.shape, np.float32)
    load().pmo_gradient_magnitude(_p(im), im.shape[0], im.shape[1], _p(g))
    return g


def resize_linear_u8(src, drows, dcols):
    """cv::resize(src, Size(dcols, drows)) with INTER_LINEAR on an 8-bit gray image (oracle/pm_oracle.h)."""
    src = c_u8(src)
    dst = np.empty((drows, dcols), np.uint8)
    load().pmo_resize_linear_u8(_p(src), src.shape[0], src.shape[1], _p(dst), drows, dcols)
    return dst


def foreground_texture_mask(gray, ksize, min_grad, downsize):
    gray = c_u8(gray)
    out = np.empty(gray.shape, np.uint8)
    lib = load()
    lib.pmo_foreground_texture_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    lib.pmo_foreground_texture_mask.restype = C.c_int
    rc = lib.pmo_foreground_texture_mask(_p(gray), gray.shape[0], gray.shape[1], ksize, min_grad, downsize, _p(out))
    if rc != 0:
        raise ValueError("the reference CHECK-fails on these arguments")
    return out


def dilate_rect(src, k):
    src = c_f32(src)
    dst = np.empty_like(src)
    load().pmo_dilate_rect(_p(src), _p(dst), src.shape[0], src.shape[1], k)
    return dst


def get_rect_subpix(src, pw, ph, cx, cy):
    if src.dtype == np.uint8:
        src = c_u8(src)
        dst = np.empty((ph, pw), np.uint8)
        load().pmo_get_rect_subpix_u8(_p(src), src.shape[0], src.shape[1], pw, ph, cx, cy, _p(dst))
    else:
        src = c_f32(src)
        dst = np.empty((ph, pw), np.float32)
        load().pmo_get_rect_subpix_f32(_p(src), src.shape[0], src.shape[1], pw, ph, cx, cy, _p(dst))
    return dst


class ImageSet:
    """Keeps the numpy planes alive next to the C struct."""

    def __init__(self, il, ir, gl=None, gr=None):
        self.il, self.ir = c_u8(il), c_u8(ir)
        self.gl = c_f32(gl) if gl is not None else gradient_magnitude(self.il)
        self.gr = c_f32(gr) if gr is not None else gradient_magnitude(self.ir)
        self.c = Images(self.il.shape[0], self.il.shape[1], _p(self.il), _p(self.ir), _p(self.gl), _p(self.gr))


def cpu_functor(pl, pr, gl, gr, f=None):
    f = f or functor()
    pl, pr, gl, gr = c_u8(pl), c_u8(pr), c_f32(gl), c_f32(gr)
    return float(load().pmo_cpu_functor(_p(pl), _p(pr), _p(gl), _p(gr), pl.size, C.byref(f)))


def cpu_cost(ims, pw, ph, x, y, d, literal=True, f=None):
    f = f or functor()
    if literal:
        return float(load().pmo_cpu_cost_literal(C.byref(ims.c), pw, ph, float(x), float(y), d, C.byref(f)))
    return float(load().pmo_cpu_cost_direct(C.byref(ims.c), pw, ph, int(x), int(y), d, C.byref(f)))


def cpu_add_noise(disp, amount, mask=None, seed=123):
    d = np.array(disp, np.float32, order="C", copy=True)
    m = c_u8(mask) if mask is not None else None
    load().pmo_cpu_add_noise(_p(d), d.shape[0], d.shape[1], amount, _p(m) if m is not None else None, seed)
    return d


def cpu_propagate(ims, disp, ph, pw, pass_mask=15, literal=False, nthreads=1, f=None):
    f = f or functor()
    d = np.array(disp, np.float32, order="C", copy=True)
    load().pmo_cpu_propagate(C.byref(ims.c), _p(d), ph, pw, C.byref(f), pass_mask, int(literal), nthreads)
    return d


def cpu_remove_background(ims, disp, ph, pw, factor=1.5, literal=False, nthreads=1, f=None):
    f = f or functor()
    d = np.array(disp, np.float32, order="C", copy=True)
    load().pmo_cpu_remove_background(C.byref(ims.c), _p(d), ph, pw, C.byref(f), factor, int(literal), nthreads)
    return d


def gpu_get_subpixel(im, row, col):
    im = c_f32(im)
    return float(load().pmo_gpu_get_subpixel(_p(im), im.shape[0], im.shape[1], row, col))


def gpu_cost5(ims, yl, xl, yr, xr, alpha=0.9):
    return float(load().pmo_gpu_cost5(C.byref(ims.c), yl, xl, yr, xr, alpha))


def gpu_add_foreground_noise(disp, unit, scale):
    d = np.array(disp, np.float32, order="C", copy=True)
    u = c_f32(unit)
    load().pmo_gpu_add_foreground_noise(_p(d), _p(u), d.size, scale)
    return d


def gpu_propagate(ims, disp, pass_mask=15, alpha=0.9, nthreads=1):
    """The Row(+1), Col(+1), Row(-1), Col(-1) sequence of patchmatch_gpu.cu:397-403."""
    d = np.array(disp, np.float32, order="C", copy=True)
    lib = load()
    if pass_mask & 1:
        lib.pmo_gpu_propagate_row(C.byref(ims.c), _p(d), 1, 3, alpha, nthreads)
    if pass_mask & 2:
        lib.pmo_gpu_propagate_col(C.byref(ims.c), _p(d), 1, 3, alpha, nthreads)
    if pass_mask & 4:
        lib.pmo_gpu_propagate_row(C.byref(ims.c), _p(d), -1, 3, alpha, nthreads)
    if pass_mask & 8:
        lib.pmo_gpu_propagate_col(C.byref(ims.c), _p(d), -1, 3, alpha, nthreads)
    return d


def gpu_mask_background(ims, disp, alpha=0.9, improve=0.8, nthreads=1):
    d = np.array(disp, np.float32, order="C", copy=True)
    load().pmo_gpu_mask_background(C.byref(ims.c), _p(d), 3, alpha, improve, nthreads)
    return d


def gpu_mask_occlusions(displ, dispr):
    dl = np.array(displ, np.float32, order="C", copy=True)
    dr = c_f32(dispr)
    load().pmo_gpu_mask_occlusions(_p(dl), _p(dr), dl.shape[0], dl.shape[1])
    return dl


def match_view(params, ims, seed):
    d = np.array(seed, np.float32, order="C", copy=True)
    load().pmo_match_view(C.byref(params), C.byref(ims.c), _p(d))
    return d


def match(params, left, right, seed_l=None, seed_r=None):
    left, right = c_u8(left), c_u8(right)
    rows, cols = left.shape
    sl = c_f32(seed_l) if seed_l is not None else None
    sr = c_f32(seed_r) if seed_r is not None else None
    dl = np.zeros((rows, cols), np.float32)
    dr = np.zeros((rows, cols), np.float32)
    load().pmo_match(C.byref(params), _p(left), _p(right), rows, cols, _p(sl) if sl is not None else None,
                     _p(sr) if sr is not None else None, _p(dl), _p(dr))
    return dl, (dr if params.left_right_check else None)


def seed_params(**kw):
    p = SeedParams()
    load().pmo_seed_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def min_eig_map(img, block_size=5):
    img = c_u8(img)
    out = np.empty(img.shape, np.float32)
    load().pmo_min_eig_map(_p(img), img.shape[0], img.shape[1], block_size, _p(out))
    return out


def corner_response_map(img, block_size=5, use_harris=0, harris_k=0.04):
    img = c_u8(img)
    out = np.empty(img.shape, np.float32)
    lib = load()
    lib.pmo_corner_response_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    lib.pmo_corner_response_map.restype = None
    lib.pmo_corner_response_map(_p(img), img.shape[0], img.shape[1], block_size, use_harris, harris_k, _p(out))
    return out


def corner_subpix(img, xs, ys, win=10, zero_zone=-1, max_iters=10, eps=0.01):
    img = c_u8(img)
    xs = np.ascontiguousarray(xs, dtype=np.float32).copy()
    ys = np.ascontiguousarray(ys, dtype=np.float32).copy()
    lib = load()
    lib.pmo_corner_subpix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_double]
    lib.pmo_corner_subpix.restype = None
    lib.pmo_corner_subpix(_p(img), img.shape[0], img.shape[1], _p(xs), _p(ys), len(xs), win, zero_zone, max_iters, eps)
    return xs, ys


def gftt_detect(img, sp=None):
    sp = sp or seed_params()
    img = c_u8(img)
    cap = max(1, sp.max_features)
    xs, ys = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    n = load().pmo_gftt_detect(_p(img), img.shape[0], img.shape[1], C.byref(sp), _p(xs), _p(ys), cap)
    return xs[:n].copy(), ys[:n].copy()


def match_rectified(left, right, kx, ky, sp=None):
    sp = sp or seed_params()
    left, right = c_u8(left), c_u8(right)
    return float(load().pmo_match_rectified(_p(left), _p(right), left.shape[0], left.shape[1], kx, ky, C.byref(sp)))


def sparse_init(left, right, dilate_factor=4, sp=None):
    sp = sp or seed_params()
    left, right = c_u8(left), c_u8(right)
    out = np.zeros(left.shape, np.float32)
    load().pmo_sparse_init(_p(left), _p(right), left.shape[0], left.shape[1], dilate_factor, C.byref(sp), _p(out))
    return out


# ---- imaging rows: oracle/pm_imaging_oracle.c -------------------------------------------------
def cpu_initialize(left, right, downsample_factor=1, sp=None):
    """Patchmatch::Initialize (patchmatch.cpp:52-87)."""
    lib = load()
    lib.pmo_cpu_initialize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(SeedParams),
                                       C.c_void_p]
    lib.pmo_cpu_initialize.restype = None
    left, right = c_u8(left), c_u8(right)
    rows, cols = left.shape
    sp = sp or seed_params()
    f = downsample_factor
    out = np.zeros((rows // f, cols // f), np.float32)
    lib.pmo_cpu_initialize(_p(left), _p(right), rows, cols, f, C.byref(sp), _p(out))
    return out


def _img_lib():
    lib = load()
    if not getattr(lib, "_img_ready", False):
        vp = C.c_void_p
        lib.pmo_disp_to_range.argtypes = [vp, C.c_size_t, C.c_double, C.c_double, vp]
        lib.pmo_disp_to_range.restype = None
        lib.pmo_remove_backscatter.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
        lib.pmo_remove_backscatter.restype = None
        lib.pmo_correct_attenuation.argtypes = [vp, vp, C.c_size_t, vp, vp]
        lib.pmo_correct_attenuation.restype = None
        lib.pmo_compute_intensity.argtypes = [vp, C.c_size_t, vp]
        lib.pmo_compute_intensity.restype = None
        lib.pmo_find_dark.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp]
        lib.pmo_find_dark.restype = C.c_float
        lib._img_ready = True
    return lib


def disp_to_range(disp, fx, baseline):
    disp = c_f32(disp)
    out = np.empty_like(disp)
    _img_lib().pmo_disp_to_range(_p(disp), disp.size, fx, baseline, _p(out))
    return out


def remove_backscatter(bgr, rng, B, beta_B):
    bgr, rng = c_f32(bgr), c_f32(rng)
    B, beta_B = c_f32(np.asarray(B, np.float32)), c_f32(np.asarray(beta_B, np.float32))
    out = np.empty_like(bgr)
    _img_lib().pmo_remove_backscatter(_p(bgr), _p(rng), rng.size, _p(B), _p(beta_B), _p(out))
    return out


def correct_attenuation(bgr, rng, X):
    bgr, rng = c_f32(bgr), c_f32(rng)
    X = c_f32(np.asarray(X, np.float32))
    out = np.empty_like(bgr)
    _img_lib().pmo_correct_attenuation(_p(bgr), _p(rng), rng.size, _p(X), _p(out))
    return out


def compute_intensity(bgr):
    bgr = c_f32(bgr)
    out = np.empty(bgr.shape[:2], np.float32)
    _img_lib().pmo_compute_intensity(_p(bgr), out.size, _p(out))
    return out


def find_dark(intensity, rng, percentile):
    intensity, rng = c_f32(intensity), c_f32(rng)
    mask = np.empty(intensity.shape, np.uint8)
    thr = _img_lib().pmo_find_dark(_p(intensity), _p(rng), intensity.shape[0], intensity.shape[1], percentile, _p(mask))
    return float(thr), mask


# ---- stereo-ready enhancement: oracle/pm_enhance_oracle.c ----------------------------------------
def _enh_lib():
    lib = load()
    if not getattr(lib, "_enh_ready", False):
        vp = C.c_void_p
        lib.pmo_gaussian_kernel.argtypes = [C.c_int, C.c_double, vp]
        lib.pmo_gaussian_blur.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp]
        lib.pmo_normalize.argtypes = [vp, C.c_int, C.c_int, vp]
        lib.pmo_normalize_color_illuminant.argtypes = [vp, C.c_int, C.c_int, vp]
        lib.pmo_stereo_ready.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        lib.pmo_value_minmax_eighth.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for n in ("pmo_gaussian_kernel", "pmo_gaussian_blur", "pmo_normalize", "pmo_normalize_color_illuminant",
                  "pmo_stereo_ready", "pmo_value_minmax_eighth"):
            getattr(lib, n).restype = None
        lib._enh_ready = True
    return lib


def gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    _enh_lib().pmo_gaussian_kernel(n, sigma, _p(k))
    return k


def gaussian_blur(src, ksize, sigma):
    src = c_f32(src)
    rows, cols = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    out = np.empty_like(src)
    _enh_lib().pmo_gaussian_blur(_p(src), rows, cols, ch, ksize, sigma, _p(out))
    return out


def normalize(bgr):
    bgr = c_f32(bgr)
    out = np.empty_like(bgr)
    _enh_lib().pmo_normalize(_p(bgr), bgr.shape[0], bgr.shape[1], _p(out))
    return out


def value_minmax_eighth(V):
    V = c_f32(V)
    lo, hi = C.c_double(0), C.c_double(0)
    _enh_lib().pmo_value_minmax_eighth(_p(V), V.shape[0], V.shape[1], C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def stereo_ready(bgr8):
    bgr8 = c_u8(bgr8)
    rows, cols = bgr8.shape[:2]
    J = np.empty((rows, cols, 3), np.float32)
    gray = np.empty((rows, cols), np.uint8)
    _enh_lib().pmo_stereo_ready(_p(bgr8), rows, cols, _p(J), _p(gray))
    return J, gray


# ---- PM_MODE_PLANES definition (oracle/pm_planes_oracle.c) ----------------------------------------------------
PMO_PL_MAX_ITERS = 16


class PlanesParams(C.Structure):
    _fields_ = [
        ("n_iters", C.c_int), ("patch", C.c_int), ("max_disp", C.c_int), ("refine_steps", C.c_int),
        ("refine_amp", C.c_float * PMO_PL_MAX_ITERS),
        ("slope_max", C.c_float), ("slope_init", C.c_float), ("slope_per_disp", C.c_float),
        ("alpha", C.c_float), ("tau_color", C.c_float), ("tau_grad", C.c_float),
        ("seed", C.c_uint64), ("left_right_check", C.c_int), ("lr_tol", C.c_float), ("state_f16", C.c_int),
        ("nthreads", C.c_int), ("window", C.c_int), ("neighbours", C.c_int),
    ]


PL_WINDOW_FULL, PL_WINDOW_CHECKER, PL_WINDOW_EVEN_COLS = 0, 1, 2
PL_NEIGH_FOUR, PL_NEIGH_TWO = 0, 1


class PlanesView(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("ref8", C.c_void_p), ("refg8", C.c_void_p),
                ("tgt8", C.c_void_p), ("tgtg8", C.c_void_p)]


class PlanesState(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("z", C.c_void_p), ("cost", C.c_void_p)]


_planes_ready = False


def _planes_lib():
    global _planes_ready
    lib = load()
    if not _planes_ready:
        vp = C.c_void_p
        PP, PV, PS = C.POINTER(PlanesParams), C.POINTER(PlanesView), C.POINTER(PlanesState)
        lib.pmo_planes_params_default.argtypes = [PP]
        lib.pmo_planes_rand.argtypes = [C.c_uint64] + [C.c_int] * 7
        lib.pmo_planes_rand.restype = C.c_uint32
        lib.pmo_planes_quant_f16.argtypes = [C.c_float]
        lib.pmo_planes_quant_f16.restype = C.c_float
        lib.pmo_planes_cost.argtypes = [PP, PV, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        lib.pmo_planes_cost.restype = C.c_float
        lib.pmo_planes_init.argtypes = [PP, PV, C.c_int, vp, PS]
        lib.pmo_planes_spatial.argtypes = [PP, PV, PS, C.c_int]
        lib.pmo_planes_view_prop.argtypes = [PP, PV, PS, PS]
        lib.pmo_planes_refine.argtypes = [PP, PV, C.c_int, C.c_int, PS]
        lib.pmo_planes_match.argtypes = [PP, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        lib.pmo_planes_prepare.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        _planes_ready = True
    return lib


def planes_params(**kw):
    p = PlanesParams()
    _planes_lib().pmo_planes_params_default(C.byref(p))
    amp = kw.pop("refine_amp", None)
    if amp is not None:
        for i, v in enumerate(amp):
            p.refine_amp[i] = v
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def planes_kwargs_of(pm_params, nthreads=8):
    """The engine's parameter struct (pm_params of include/pm/patchmatch.h, ctypes) -> keyword arguments of
    planes_params(): the SAME plane-mode configuration for the CPU definition."""
    q = pm_params
    return dict(n_iters=q.patchmatch_iters, patch=q.patch_w[0], max_disp=q.max_disp, refine_steps=q.plane_refine_steps,
                refine_amp=[q.noise_amp[i] for i in range(16)], slope_max=q.plane_slope_max,
                slope_init=q.plane_slope_init, slope_per_disp=q.plane_slope_per_disp, alpha=q.functor_alpha,
                tau_color=q.functor_tau_color, tau_grad=q.functor_tau_grad, seed=q.noise_seed,
                left_right_check=q.left_right_check, lr_tol=q.plane_lr_tol, state_f16=q.state_dtype,
                window=q.plane_window, neighbours=q.plane_neighbours, nthreads=nthreads)


def planes_rand(seed, stage, it, k, view, draw, x, y):
    return int(_planes_lib().pmo_planes_rand(seed, stage, it, k, view, draw, x, y))


def planes_quant_f16(v):
    return float(_planes_lib().pmo_planes_quant_f16(float(v)))


class PlanesViews:
    """The u8 planes of both views of a pair + state arrays; keeps every buffer alive."""

    def __init__(self, left, right):
        left, right = c_u8(left), c_u8(right)
        self.rows, self.cols = left.shape
        n = left.size
        self.buf = [np.zeros((4, self.rows, self.cols), np.uint8) for _ in range(2)]
        _planes_lib().pmo_planes_prepare(_p(left), _p(right), self.rows, self.cols, _p(self.buf[0]), _p(self.buf[1]))
        self.view = []
        for v in range(2):
            b = self.buf[v]
            self.view.append(PlanesView(self.rows, self.cols, b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data,
                                        b[3].ctypes.data))
        self.planes = [np.zeros((4, self.rows, self.cols), np.float32) for _ in range(2)]
        self.state = [PlanesState(*[self.planes[v][k].ctypes.data for k in range(4)]) for v in range(2)]

    def cost(self, p, view, x, y, a, b, z):
        return float(_planes_lib().pmo_planes_cost(C.byref(p), C.byref(self.view[view]), x, y, a, b, z))

    def init(self, p, view, seed=None):
        s = c_f32(seed) if seed is not None else None
        _planes_lib().pmo_planes_init(C.byref(p), C.byref(self.view[view]), view, _p(s) if s is not None else None,
                                      C.byref(self.state[view]))

    def spatial(self, p, view, parity):
        _planes_lib().pmo_planes_spatial(C.byref(p), C.byref(self.view[view]), C.byref(self.state[view]), parity)

    def view_prop(self, p, view):
        _planes_lib().pmo_planes_view_prop(C.byref(p), C.byref(self.view[view]), C.byref(self.state[view]),
                                           C.byref(self.state[1 - view]))

    def refine(self, p, view, it):
        _planes_lib().pmo_planes_refine(C.byref(p), C.byref(self.view[view]), view, it, C.byref(self.state[view]))


def planes_match(p, left, right, seed_l=None, seed_r=None, want_planes=False):
    left, right = c_u8(left), c_u8(right)
    rows, cols = left.shape
    sl = c_f32(seed_l) if seed_l is not None else None
    sr = c_f32(seed_r) if seed_r is not None else None
    dl = np.zeros((rows, cols), np.float32)
    dr = np.zeros((rows, cols), np.float32)
    nv = 2 if p.left_right_check else 1
    planes = np.zeros((nv, 4, rows, cols), np.float32) if want_planes else None
    _planes_lib().pmo_planes_match(C.byref(p), _p(left), _p(right), rows, cols, _p(sl) if sl is not None else None,
                                   _p(sr) if sr is not None else None, _p(dl), _p(dr),
                                   _p(planes) if planes is not None else None)
    return (dl, dr, planes) if want_planes else (dl, dr)
